set -e
for sh in "64 256 256 40 40 3 1" "64 512 512 20 20 3 1" "32 640 640 40 40 3 1" "32 128 128 40 40 3 1" "32 128 256 80 80 3 2" "32 256 512 40 40 3 2" "64 512 512 40 40 1 1"; do
  echo "== $sh"; timeout -k 10 100 python tools/bench_conv.py $sh 20 all 2>/dev/null | sort -t' ' -k4 -n | head -8
done
