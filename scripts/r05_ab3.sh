# usage (GPU box): bash scripts/r05_ab3.sh <tag>  -- CompactKernel's LDS image size (ids a wave holds; 0 = search form) on C2
TAG=${1:-r05z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
for S in 2048 4096 1024 0; do
  SPMX_COMPACT_STAGED=$S timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs > $O/bench_$S.json 2> $O/bench_$S.err
  DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/kernel_stats_$S.txt 2>&1; rm -rf $O/prof
  echo "== image $S ids"; grep -E "CompactKernel|ScanFinal" $O/kernel_stats_$S.txt | cut -c1-150
  python -c "
import json,sys
d=json.loads(open('$O/bench_$S.json').read().strip().splitlines()[-1]); print('%.4g sentences/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
