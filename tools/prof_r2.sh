set -e
mkdir -p gpurun_out/r2f && cd /root/repo
export TMPDIR=/tmp Y5_TUNE_CACHE=$PWD/gpurun_out/r2f/tune_s.json
timeout -k 10 200 python tools/profile_ops.py > gpurun_out/r2f/per_op_s.txt 2>&1
timeout -k 10 200 python bench.py --steps 30 > gpurun_out/r2f/bench_a.json 2> gpurun_out/r2f/bench_a.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2f/trace -o s -- python bench.py --steps 50 --warmup 5 > gpurun_out/r2f/bench_prof.json 2> gpurun_out/r2f/bench_prof.err
ls -R gpurun_out/r2f/trace | head -20
