# usage (GPU box): bash scripts/r05_first.sh <tag> -- first look at the word-per-lane rounds (kernels_wordwave.h): the GPU tests of
# the word form, then the C2 line with the new form in both rounds / the first only / neither (SPMX_WORD_WAVE = 3 / 1 / 0)
TAG=${1:-r05a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests/test_word_form.py tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_gpu_word.txt 2>&1; tail -4 $O/pytest_gpu_word.txt
for W in 3 1 0; do
  SPMX_WORD_WAVE=$W timeout 600 python bench.py --no-cpu-baseline --no-side-configs --steps 5 --warmup 2 > $O/bench_w$W.json 2> $O/bench_w$W.err
  python - $O/bench_w$W.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], str(d.get("probe_ids_bit_exact"))[:160], json.dumps(d["roofline"].get("all_kernels_ms"))[:600], "alone", json.dumps(d["roofline"].get("alone"))[:200])
lp = d.get("long_piece_model") or {}
print("  w16", lp.get("value"), lp.get("ms_per_step"), json.dumps(lp.get("kernels_ms", {}))[:400])
PY
done
for WV in 8 12; do
  SPMX_WORDWAVE_WAVES=$WV timeout 600 python bench.py --no-cpu-baseline --no-side-configs --no-second-model --steps 5 --warmup 2 > $O/bench_wv$WV.json 2> $O/bench_wv$WV.err
  python -c "
import json,sys
d=json.loads(open('$O/bench_wv$WV.json').read().strip().splitlines()[-1]); print('waves $WV', d['value'], d['ms_per_step'], json.dumps(d['roofline'].get('all_kernels_ms'))[:500])"
done
ls $O
