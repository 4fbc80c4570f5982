# usage (GPU box): bash scripts/r04_exp10.sh -- the full GPU suite on the head, the window-reuse word loop on C2, C5 short form.  gpurun_out/r04l/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04l; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs"
timeout 200 $B > $O/bench_head.json 2> $O/bench_head.err
timeout 200 $B --model c5_250k --sentences 1000000 --steps 20 --warmup 18 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 200 $B --model bpe32k > $O/bench_bpe.json 2> $O/bench_bpe.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04l/bench_head.json"))
print("head %.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], "| w16 %.3f ms" % d["long_piece_model"]["ms_per_step"], d["long_piece_model"]["kernels_ms"])
for m in ("c5", "bpe"):
    d = json.load(open("gpurun_out/r04l/bench_%s.json" % m))
    print(m, "%.3f ms/step %.1f M/s" % (d["ms_per_step"], d["value"] / 1e6), d["roofline"]["all_kernels_ms"])
PY
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
