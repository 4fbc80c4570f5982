# usage (GPU box): bash scripts/r03_ab_model.sh MODEL [SENTENCES] "ENV=1" ...  -- one short bench of MODEL per environment string
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M=$1; N=$2; shift; shift
for e in "$@"; do
  env $e python bench.py --model $M --sentences $N --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$M $e', '%.1f M/s' % (d['value']/1e6), '%.3f ms' % d['ms_per_step'], d['roofline']['all_kernels_ms'], {k: round(v / 1e9, 2) for k, v in (d['roofline'].get('phase_cycles') or {}).items()}, d['roofline'].get('path'))"
done
