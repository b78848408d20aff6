#!/bin/bash
# Hardware-counter passes over the default bench workload, one rocprofv3 run per counter set
# (the SQ / TCC blocks have few counter slots; asking for too many in one run hangs the profiler).
# usage: scripts/pmc_bench.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/<pass>/
set -e
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
run_pass() {
	name=$1; shift
	timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-host-buffers $BENCH_ARGS > $out/$name.json 2> $out/$name.err
}
BENCH_ARGS="$*"
run_pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR
run_pass fetch FETCH_SIZE
run_pass l2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
ls $out
