# usage (GPU box): bash scripts/r02_check.sh <tag>  -- GPU parity tests + headline benches after a kernel change
TAG=${1:-r02a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 600 $B > $O/uni.json 2> $O/uni.err
timeout 600 $B --model bpe32k > $O/bpe.json 2> $O/bpe.err
timeout 600 $B --model c5_250k --sentences 1000000 > $O/c5.json 2> $O/c5.err
timeout 600 $B --unsorted > $O/uni_unsorted.json 2> $O/uni_unsorted.err
for f in $O/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print(d.get('value'), d.get('ms_per_step'), r.get('all_kernels_ms'), r.get('phase_cycles'))
except Exception as e: print('ERR',e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
