# usage (GPU box): bash scripts/r02_explore.sh   -- round-2 baseline exploration: occupancy sensitivity, forced ring 32, n-best rate
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_explore; mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 600 $B > $O/base.json 2> $O/base.err
for w in 6 8 10; do SPMX_TILE_WAVES=$w timeout 300 $B > $O/waves$w.json 2>> $O/ab.err; done
SPMX_FORCE_RING=32 timeout 300 $B > $O/ring32.json 2>> $O/ab.err
SPMX_SUB_BUCKETS=64 timeout 300 $B > $O/sub64.json 2>> $O/ab.err
timeout 300 python scripts/nbest_rate.py 200000 5 > $O/nbest5.json 2>> $O/side.err
timeout 300 python scripts/nbest_rate.py 50000 64 > $O/nbest64.json 2>> $O/side.err
for f in $O/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print(d.get('value'), d.get('ms_per_step'), r.get('all_kernels_ms'), r.get('phase_cycles'))
except Exception as e: print('ERR',e)
PY
done
