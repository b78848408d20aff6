#!/bin/bash
# usage: [BENCH_ARGS="--layout indexed --bvh 'sah 16'"] scripts/sweep_env.sh OUTFILE "VAR=a VAR2=b" "VAR=c" ...   -- one bench.py run (traversal only) per setting
out=$1; shift
: > "$out"
for setting in "$@"; do
	line=$(eval env $setting timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu --no-host-buffers --no-extra $BENCH_ARGS 2>/dev/null | grep '^{') || exit 1
	echo "$setting $(echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['triangle_tests_per_ray'])")" | tee -a "$out"
done
