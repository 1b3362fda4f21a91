set -e
for sh in "32 32 64 320 320 3 2" "32 32 32 160 160 3 1" "32 64 64 80 80 3 1" "32 64 128 160 160 3 2" "32 128 256 80 80 3 2"; do
  echo "== $sh"; timeout -k 10 100 python tools/bench_conv.py $sh 20 all 2>/dev/null | sort -t' ' -k4 -n | head -6
done
