# usage (GPU box): bash scripts/r05_second.sh <tag> -- the pipelined word-per-lane rounds: GPU tests of the word form, the C2 line, wavefront counts
TAG=${1:-r05b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests/test_word_form.py tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_gpu_word.txt 2>&1; tail -4 $O/pytest_gpu_word.txt
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "%.4g" % d["value"], "%.3f ms" % d["ms_per_step"], json.dumps(r.get("all_kernels_ms")), "alone", json.dumps((r.get("alone") or {}).get("kernel_ms")), "trips", (r.get("phase_cycles") or {}).get("search_trips"))
lp = d.get("long_piece_model") or {}
if lp: print("  w16 %.4g %.3f ms" % (lp.get("value"), lp.get("ms_per_step")), json.dumps(lp.get("kernels_ms", {})))
PY
}
timeout 600 python bench.py --no-cpu-baseline --no-side-configs --steps 5 --warmup 2 > $O/bench_w3.json 2> $O/bench_w3.err; show $O/bench_w3.json
for WV in 8 10; do
  SPMX_WORDWAVE_WAVES=$WV timeout 600 python bench.py --no-cpu-baseline --no-side-configs --no-second-model --steps 5 --warmup 2 > $O/bench_wv$WV.json 2> $O/bench_wv$WV.err; show $O/bench_wv$WV.json
done
SPMX_WORD_WAVE=1 timeout 600 python bench.py --no-cpu-baseline --no-side-configs --steps 5 --warmup 2 > $O/bench_w1.json 2> $O/bench_w1.err; show $O/bench_w1.json
ls $O
