# usage (GPU box): bash scripts/r02_ab.sh <tag> "<ENV=..> <ENV=..>" ...   -- headline bench under each environment
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout 600 $B $BENCH_ARGS > $O/ab$i.json 2> $O/ab$i.err
  echo "== [$envs]"; python - $O/ab$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print(d.get('value'), d.get('ms_per_step'), r.get('all_kernels_ms'), r.get('phase_cycles'))
except Exception as e: print('ERR',e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
