set -e
for sh in "64 512 512 20 20 3 1" "64 256 256 40 40 3 1" "32 640 640 40 40 3 1" "32 128 128 40 40 3 1" "32 64 64 80 80 3 1" "32 256 256 20 20 3 1" "32 128 256 80 80 3 2"; do
  for ko in 0 1; do
    for v in 21 25 28 30 37 41; do
      echo -n "korder $ko v$v: "; Y5_KORDER=$ko timeout -k 10 60 python tools/bench_conv.py $sh 20 $v 2>/dev/null
    done
  done
done
