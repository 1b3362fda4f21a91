# Every compiled conv variant on a list of layer shapes, best five per shape.  usage: sweep_conv.sh ["B Cin Cout H W k s" ...]
set -e
R=$(dirname "$0")/..
if [ $# -eq 0 ]; then
  set -- "32 32 64 320 320 3 2" "32 32 32 160 160 3 1" "32 64 64 80 80 3 1" "32 128 128 40 40 3 1" "32 256 256 20 20 3 1" \
         "32 128 256 80 80 3 2" "32 256 512 40 40 3 2" "32 128 128 80 80 1 1" "32 256 256 40 40 1 1" "64 512 512 20 20 3 1" "32 640 640 40 40 3 1"
fi
for sh in "$@"; do
  echo "== $sh"; timeout -k 10 100 python $R/tools/bench_conv.py $sh 20 all 2>/dev/null | sort -t' ' -k4 -n | head -5
done
