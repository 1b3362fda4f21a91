set -e
for sh in "64 512 512 20 20 3 1" "32 128 128 40 40 3 1"; do
  for v in 25 40 41 39; do
    echo "== $sh v$v"; STAMP_KIND=igemm timeout -k 10 60 python tools/stamp_conv.py $sh $v 2>/dev/null
  done
done
