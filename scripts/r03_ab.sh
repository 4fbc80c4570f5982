# usage (GPU box): bash scripts/r03_ab.sh "ENV=1 ENV2=x" ...   -- one short bench per environment string, one line each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for e in "$@"; do
  env $e python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$e', '%.1f M/s' % (d['value']/1e6), '%.3f ms' % d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
