# Counter passes for one convolution shape / kernel variant: pmc_conv.sh NAME B Cin Cout H W k s VARIANT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
args="$1 $2 $3 $4 $5 $6 $7 4 $8"
pass() { # tag counters...
  tag=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${name}_$tag -- python3 $R/tools/bench_conv.py $args > $R/gpurun_out/pmc_${name}_$tag.log 2>&1
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU &&
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS &&
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum &&
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum &&
pass grbm GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum
for t in sq sq2 tcc tcp grbm; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${name}_$t conv; done > $R/gpurun_out/pmc_${name}.txt 2>&1
tail -60 $R/gpurun_out/pmc_${name}.txt
