set -e
for mode in 0 1 2 4 5; do
  for sh in "64 512 512 20 20 3 1" "32 640 640 40 40 3 1" "32 128 128 40 40 3 1"; do
    for v in 39 40 41 42; do
      echo -n "mode $mode v$v: "; Y5_PP_MODE=$mode timeout -k 10 60 python tools/bench_conv.py $sh 20 $v 2>/dev/null
    done
  done
done
