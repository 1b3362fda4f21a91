cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name args...
  name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/tools/bench_conv.py "$@" 4 > $R/gpurun_out/pmc_$name.log 2>&1
  timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc2_$name -- python3 $R/tools/bench_conv.py "$@" 4 > $R/gpurun_out/pmc2_$name.log 2>&1
}
run c3_64 32 64 64 80 80 3 1 && run c1_128 32 128 128 40 40 1 1 && run c1_64 32 64 64 160 160 1 1 && run c3_256 32 256 256 20 20 3 1
for f in c3_64 c1_128 c1_64 c3_256; do tail -1 $R/gpurun_out/pmc_$f.log; done
