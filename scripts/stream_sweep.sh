#!/bin/bash
# Sweeps the chunk layout of the streamed host path: prints, per layout, when each chunk was uploaded and sorted
# and the page-locked host-batch rate.  usage: scripts/stream_sweep.sh "0.2,0.5,1" "0.12,0.4,1" ...
set -o pipefail
for parts in "$@"; do
	echo "== parts $parts ${EXTRA_ENV}"
	env RTGI_HIP_STREAM_PARTS=$parts RTGI_HIP_STREAM_TIMES=1 $EXTRA_ENV timeout -k 10 120 python3 scripts/host_path_bench.py 2>&1 | tail -2 | cut -c1-420 || exit 1
done
