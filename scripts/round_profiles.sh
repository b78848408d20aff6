#!/bin/bash
# The measurements that back the numbers of a round, in one go on the GPU box: counter passes -> traffic.json, the
# kernel-trace statistics of bench.py, and the bench line itself (which then prices its floors with the fresh counters).
# usage: scripts/round_profiles.sh <tag>      results under gpurun_out/<tag>/ and gpurun_out/pmc_<tag>/
set -e -o pipefail
tag=$1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ -z "$SKIP_PMC" ]; then scripts/pmc_bench.sh $tag; fi
python3 scripts/pmc_to_traffic.py gpurun_out/pmc_$tag soup1m_bounce $out/pmc_summary.txt > $out/traffic_entry.json
cp profiles/traffic.json $out/traffic.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-host-buffers --no-extra > $out/bench_under_profiler.json 2> $out/bench_under_profiler.err
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
