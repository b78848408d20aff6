# A/B of scripts/experiments/r03_trail_more_waves.patch: libraries librtgi_hip_t{B,C,D,E}.so are built from the patched source with
# -DRTGI_TRAIL=1 and -DRTGI_BLOCK=1024|768|640|512 (-DRTGI_WAVES_PER_SIMD=4|6|5|6); numbers in profiles/r03_trail_more_waves_ab.txt
set -e
run() { # name lib bpc
  export RTGI_HIP_LIB=$2; export RTGI_HIP_BLOCKS_PER_CU=$3
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/trail_$1.smoke 2>&1 || { echo "$1 smoke FAILED"; tail -3 gpurun_out/trail_$1.smoke; return 0; }
  for i in 1 2; do
    timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu --no-host-buffers --no-extra > gpurun_out/trail_$1_$i.json 2>gpurun_out/trail_$1.err
    python -c "
import json,sys
d=json.loads(open('gpurun_out/trail_$1_$i.json').read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_ms'))"
  done
}
run A librtgi_hip.so 1
run B librtgi_hip_tB.so 1
run C librtgi_hip_tC.so 2
run D librtgi_hip_tD.so 2
run E librtgi_hip_tE.so 3
run A2 librtgi_hip.so 1
