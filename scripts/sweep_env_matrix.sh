#!/bin/bash
# usage: scripts/sweep_env_matrix.sh CASES "VAR=a" "VAR=b" ...  -- scripts/bench_matrix.py (GPU legs only) per setting; prints kernel ms per population
cases=$1; shift
for setting in "$@"; do
	env $setting timeout -k 10 300 python scripts/bench_matrix.py --cases $cases --no-cpu --out gpurun_out/sweep_matrix.jsonl > /dev/null 2>&1 || exit 1
	python - "$setting" <<'PY'
import json, sys
rows = [json.loads(l) for l in open("gpurun_out/sweep_matrix.jsonl")]
print(sys.argv[1], " ".join("%s/%s %.3f" % (r["case"], r["population"], r["kernel_ms"]) for r in rows))
PY
done
