set -e
for sh in "64 512 512 20 20 3 1" "32 128 128 40 40 3 1" "64 512 512 40 40 1 1"; do
  for v in 25 26 37 2; do
    echo "== $sh v$v"; STAGING_ONLY=1 timeout -k 10 60 python tools/stamp_conv.py $sh $v 2>/dev/null; timeout -k 10 60 python tools/stamp_conv.py $sh $v 2>/dev/null
  done
done
