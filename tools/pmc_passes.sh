# rocprofv3 counter passes over one workload (each pass its own run; counters per pass within the hardware's limits).
# usage: bash tools/pmc_passes.sh <tag> <one_dispatch args...>     results: gpurun_out/pmc_<tag>_<pass>/
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  n=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" -d $R/gpurun_out/pmc_${tag}_$n -o g --output-format csv -- python3 $R/tools/one_dispatch.py "${ARGS[@]}" > $R/gpurun_out/pmc_${tag}_$n.log 2>&1
}
ARGS=("$@")
run sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run sq2 SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_CYCLES
run ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run td TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
run mem FETCH_SIZE WRITE_SIZE TCC_HIT_sum
