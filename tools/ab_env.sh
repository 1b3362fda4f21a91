# A/B of one environment setting on ONE box (boxes differ by +-5 %): bench.py alternately without and with it.
# usage (on the GPU box): bash tools/ab_env.sh NAME=VALUE
cd $GRAFT_REPO_ROOT
for v in off on off on; do
  if [ $v = on ]; then export "$1"; else unset "${1%%=*}"; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/abenv_$v.json 2> gpurun_out/abenv_$v.err || exit 1
  python3 - $v <<'PY'
import json, sys
d = json.load(open('gpurun_out/abenv_%s.json' % sys.argv[1]))
print(sys.argv[1], 'images/s %.0f  ms/step %.4f  forward %.4f  conv %.4f' % (d['value'], d['ms_per_step'], d['stages']['forward_ms'], d['stages']['conv_ms']))
PY
done
