#!/bin/bash
# Host-buffer rate of bench.py (ordinary and page-locked memory) over chunking settings of rtgi_hip_closest_hit.
# usage: scripts/sweep_host_chunk.sh "VAR=value,VAR2=value" ...   ("-" for the defaults)
for setting in "$@"; do
	envs=()
	if [ "$setting" != "-" ]; then IFS=',' read -ra envs <<< "$setting"; fi
	echo -n "$setting: "
	env "${envs[@]}" timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -n 1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["host_buffer_path"]["value"], d["host_buffer_path"].get("page_locked"))'
done
