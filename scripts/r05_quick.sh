TAG=${1:-r05q}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "%.4g" % d["value"], "%.3f ms" % d["ms_per_step"], "pipeline %.3f" % r.get("pipeline_ms", 0), json.dumps(r.get("all_kernels_ms")))
lp = d.get("long_piece_model") or {}
if lp: print("  w16 %.4g %.3f ms" % (lp.get("value"), lp.get("ms_per_step")), json.dumps(lp.get("kernels_ms", {})), lp.get("probe_ids_bit_exact"))
PY
}
timeout 600 python bench.py --no-cpu-baseline --no-side-configs --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; show $O/bench.json
timeout 600 python bench.py --model bpe32k --no-cpu-baseline --no-side-configs --steps 4 --warmup 2 > $O/bench_bpe.json 2> $O/bench_bpe.err; show $O/bench_bpe.json
if [ -n "$WITH_TESTS" ]; then ( time timeout 900 python -m pytest tests/test_word_form.py tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_gpu_word.txt 2>&1; grep -E "passed|failed" $O/pytest_gpu_word.txt | tail -2; fi
