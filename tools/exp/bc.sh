set -e
for sh in "32 80 80 160 160 3 1" "32 48 48 160 160 3 1" "32 80 160 320 320 3 2" "32 128 128 40 40 3 1" "32 64 64 80 80 3 1" "32 32 32 160 160 3 1" "64 512 512 20 20 3 1"; do
  echo "== $sh"; timeout -k 10 100 python tools/bench_conv.py $sh 20 all 2>/dev/null | sort -t' ' -k4 -n | head -5
done
