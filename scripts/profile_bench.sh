#!/bin/bash
# Kernel-trace profile of the default bench workload (run on the GPU box through gpurun).
# usage: scripts/profile_bench.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-host-buffers "$@" > $out/bench.json 2> $out/bench.err
ls -R $out | head -30
