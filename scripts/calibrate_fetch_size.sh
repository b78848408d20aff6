#!/bin/bash
# Calibrates FETCH_SIZE for this kernel's access pattern (random 64-byte records, four 16-byte loads per lane):
# runs the gather micro-benchmark under rocprofv3 --pmc FETCH_SIZE; the byte count is known from the record count.
set -e
out=gpurun_out/calib_fetch
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o pmc -- scripts/ubench/gather_quads > $out/ubench.txt 2> $out/ubench.err
cat $out/ubench.txt
