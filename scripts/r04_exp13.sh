# usage (GPU box): bash scripts/r04_exp13.sh -- the one-round-trip memo lookup on w16 / C3 / the open-vocabulary corpus.  gpurun_out/r04o/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04o; mkdir -p $O
( time timeout 600 python -m pytest tests/test_word_form.py -m gpu -x -q ) > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04o/bench.json"))
print("head %.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"])
for k in ("long_piece_model", "c3", "natural_open_vocab", "natural_botchan_x2000", "c5"):
    v = d.get(k, {})
    print(k, "%.3f ms %.1f M/s" % (v.get("ms_per_step", 0), v.get("value", 0) / 1e6), v.get("kernels_ms"), v.get("probe_ids_bit_exact"))
PY
