# Round profile: kernel trace + stats of the bench command on the TUNED plan only (variant choices are made by an
# untraced run first and replayed from Y5_TUNE_CACHE), then the two HBM counter passes and an MFMA-busy pass, joined per
# launch by tools/roofline_table.py.
# The traced and counter runs issue ONE step at a time (--in-flight 1): launch i of the trace is op i of the plan; the first,
# untraced run is the bench as the driver runs it (several steps in flight).
# usage: round_profile.sh TAG [MODEL BATCH [key=value ...]]     (key=value pairs are copied into the roofline record)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; model=${2:-yolov5s}; batch=${3:-32}; shift; shift; shift
B="--model $model --batch $batch"
export Y5_TUNE_CACHE=$R/gpurun_out/tune_$tag.json
timeout -k 10 400 python3 $R/bench.py $B --steps 20 --warmup 5 --ops-json $R/gpurun_out/ops_$tag.json > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py $B --steps 20 --warmup 5 --no-cpu-baseline --no-latency --in-flight 1 > $R/gpurun_out/prof_$tag.log 2>&1 &&
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$tag -- python3 $R/bench.py $B --steps 3 --warmup 1 --no-cpu-baseline --no-latency --in-flight 1 > $R/gpurun_out/pmc_fetch_$tag.log 2>&1 &&
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$tag -- python3 $R/bench.py $B --steps 3 --warmup 1 --no-cpu-baseline --no-latency --in-flight 1 > $R/gpurun_out/pmc_write_$tag.log 2>&1 &&
timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma_$tag -- python3 $R/bench.py $B --steps 3 --warmup 1 --no-cpu-baseline --no-latency --in-flight 1 > $R/gpurun_out/pmc_mfma_$tag.log 2>&1 &&
python3 $R/tools/mfma_busy.py $R/gpurun_out/pmc_mfma_$tag > $R/gpurun_out/mfma_busy_$tag.txt &&
python3 $R/tools/hbm_traffic.py $R/gpurun_out/pmc_fetch_$tag $R/gpurun_out/pmc_write_$tag $R/gpurun_out/hbm_traffic_$tag.json $model $batch > /dev/null &&
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) && cp $f $R/gpurun_out/kernel_stats_$tag.csv &&
t=$(find $R/gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) && python3 $R/tools/trace_layers.py $t > $R/gpurun_out/trace_layers_$tag.txt && tail -3 $R/gpurun_out/trace_layers_$tag.txt &&
python3 $R/tools/roofline_table.py $R/gpurun_out/ops_$tag.json $t $R/gpurun_out/pmc_fetch_$tag $R/gpurun_out/pmc_write_$tag $R/gpurun_out/pmc_mfma_$tag $R/gpurun_out/roofline_$tag.json tag=$tag "$@" > $R/gpurun_out/roofline_$tag.txt && cat $R/gpurun_out/roofline_$tag.txt
