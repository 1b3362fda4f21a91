# A/B of two builds of the library on ONE box (boxes differ by +-5 %): ab/liby5hip_old.so against ab/liby5hip_new.so,
# alternating, bench.py each time.  usage (on the GPU box): bash tools/ab_lib.sh
cd $GRAFT_REPO_ROOT
for v in old new old new; do
  cp ab/liby5hip_$v.so yolo_amd/liby5hip.so &&
  timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python3 - $v <<'PY'
import json, sys
d = json.load(open('gpurun_out/ab_%s.json' % sys.argv[1]))
print(sys.argv[1], 'images/s %.0f  ms/step %.4f  forward %.4f  plan %.4f' % (d['value'], d['ms_per_step'], d['stages']['forward_ms'], d['stages']['plan_ms']))
PY
done
cp ab/liby5hip_new.so yolo_amd/liby5hip.so
