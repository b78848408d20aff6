#!/bin/bash
# Hardware-counter passes over the default bench workload, one rocprofv3 run per counter set
# (every block has few counter slots: 8 SQ, 4 TCC, 4 TCP, 2 TA, 2 TD, 2 GRBM; asking for more in one run hangs
# the profiler).
# usage: scripts/pmc_bench.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/<pass>/
#        PASSES="ta1 tcp1" scripts/pmc_bench.sh <tag>   runs only the named passes
set -e
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH_ARGS="$*"
run_pass() {
	name=$1; shift
	if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $name "; then return 0; fi
	timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-host-buffers --no-extra $BENCH_ARGS > $out/$name.json 2> $out/$name.err
	echo "pass $name done"
}
# instruction mix, occupancy, waiting
run_pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR
run_pass sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
# memory traffic beyond L2, L2 hit rate
run_pass fetch FETCH_SIZE
run_pass l2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
# the vector memory pipeline itself: texture addresser (TA), vector cache (TCP), texture data return (TD)
run_pass ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run_pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run_pass ta3 TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum
run_pass ta4 TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_COALESCED_READ_CYCLES_sum
run_pass ta5 TA_BUSY_avr TA_BUSY_max
run_pass tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run_pass tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run_pass tcp3 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
run_pass tcp4 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum
run_pass td1 TD_TD_BUSY_sum TD_TC_STALL_sum
run_pass td2 TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum
ls $out
