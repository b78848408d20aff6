#!/bin/bash
# A/B runs of bench.py inside one GPU call: each argument is a comma-separated list of VAR=value settings
# ("-" for none); prints workload value and kernel time per setting.
# usage: scripts/ab_bench.sh "RTGI_HIP_BLOCKS_PER_CU=3" "RTGI_HIP_BLOCKS_PER_CU=4,RTGI_HIP_COMPACT_NODES=0" ...
for setting in "$@"; do
	envs=()
	if [ "$setting" != "-" ]; then IFS=',' read -ra envs <<< "$setting"; fi
	out=$(env "${envs[@]}" timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu --no-host-buffers $BENCH_ARGS 2>/dev/null | tail -n 1)
	echo "$setting: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s, kernel", d["roofline"]["kernel_ms"], "ms, step", d["ms_per_step"], "ms")')"
done
