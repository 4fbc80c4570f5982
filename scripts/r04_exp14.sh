# usage (GPU box): bash scripts/r04_exp14.sh -- the call-local table sized by the last call (host-side change).  gpurun_out/r04p/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; mkdir -p $O
( time timeout 600 python -m pytest tests/test_word_form.py -m gpu -x -q ) > $O/tests.log 2>&1; tail -3 $O/tests.log
( time timeout 900 python bench.py > $O/bench_uni32k_10m.json 2> $O/bench.err ) 2> $O/bench_wall.txt; tail -3 $O/bench_wall.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04p/bench_uni32k_10m.json"))
print("head %.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], d["roofline"]["traffic"] is not None)
for k in ("long_piece_model", "c3", "natural_open_vocab", "natural_botchan_x2000", "c5"):
    v = d.get(k, {})
    print(k, "%.3f ms %.1f M/s" % (v.get("ms_per_step", 0), v.get("value", 0) / 1e6), v.get("kernels_ms"), v.get("probe_ids_bit_exact"))
print(d["end_to_end"]["value"], d["end_to_end"]["best"])
PY
