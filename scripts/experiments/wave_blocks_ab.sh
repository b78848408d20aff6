# per-wave block claims (scripts/experiments/r03_wave_blocks.patch applied, built with -DRTGI_WAVE_BLOCKS=128|256|512 as librtgi_hip_wb*.so)
# A/B of per-wave block claims (-DRTGI_WAVE_BLOCKS=128|256|512 builds as librtgi_hip_wb*.so) against the shipped library
set -e
for lib in librtgi_hip.so librtgi_hip_wb128.so librtgi_hip_wb256.so librtgi_hip_wb512.so librtgi_hip.so; do
  RTGI_HIP_LIB=$lib timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > /dev/null 2>&1 || { echo "$lib smoke FAILED"; continue; }
  RTGI_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 40 --warmup 10 --no-cpu --no-host-buffers --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
