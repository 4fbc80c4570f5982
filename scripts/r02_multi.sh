# usage (GPU box): bash scripts/r02_multi.sh <tag> "<bench args>" ...   -- bench.py once per argument string
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
i=0
for a in "$@"; do
  i=$((i+1))
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $a > $O/m$i.json 2> $O/m$i.err
  echo "== [$a]"; python - $O/m$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print(d.get('value'), d.get('ms_per_step'), r.get('all_kernels_ms'), r.get('phase_cycles'))
except Exception as e: print('ERR',e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
