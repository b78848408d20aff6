#!/bin/bash
# A/B of library builds inside one GPU call: scripts/ab_bench.sh "<lib>[:ENV=V,...]" ...  (each run twice, interleaved)
# prints Mrays/s, ms per step, kernel ms, share of inner visits served from LDS
set -e
for round in 1 2; do
	for spec in "$@"; do
		lib=${spec%%:*}
		envs=""
		if [ "$spec" != "$lib" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
		env RTGI_HIP_LIB=$lib $envs timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu --no-host-buffers --no-extra $BENCH_ARGS 2>/dev/null |
			python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-60s %8.2f Mrays/s  step %.4f ms  kernel %.4f ms  lds %.4f' % ('$spec', d['value'], d['ms_per_step'], r['kernel_ms'], r['inner_visits_served_from_lds']))"
	done
done
