# Memory-path counters for one convolution: pmc_mem.sh NAME B Cin Cout H W k s VARIANT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
args="$1 $2 $3 $4 $5 $6 $7 4 $8"
pass() { tag=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pm_${name}_$tag -- python3 $R/tools/bench_conv.py $args > $R/gpurun_out/pm_${name}_$tag.log 2>&1
}
pass a TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_CYCLE_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE &&
pass b TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum TCC_READ_sum TCC_HIT_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum
for t in a b; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pm_${name}_$t conv; done > $R/gpurun_out/pm_${name}.txt 2>&1
cat $R/gpurun_out/pm_${name}.txt
