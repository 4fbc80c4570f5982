# usage (GPU box): bash scripts/r04_exp3.sh -- round 4, third session: the TX word kernels (text staged in LDS) against
# the round-3 forms, at a few workgroup widths; parity first.  Results under gpurun_out/r04d/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04d; mkdir -p $O
( time timeout 600 python -m pytest tests/test_word_form.py -m gpu -x -q --durations=5 ) > $O/wordform.log 2>&1; tail -4 $O/wordform.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs"
SPMX_WORD_TX=0 timeout 200 $B > $O/bench_notx.json 2> $O/bench_notx.err
for v in 8 9 6 4; do
  SPMX_WORD_TX_WAVES=$v timeout 200 $B > $O/bench_tx$v.json 2> $O/bench_tx$v.err
done
SPMX_NO_OVERLAP=1 timeout 200 $B > $O/bench_tx8_noov.json 2> $O/bench_tx8_noov.err
python - <<'PY'
import json
for v in ("notx", "tx8", "tx9", "tx6", "tx4", "tx8_noov"):
    try:
        d = json.load(open("gpurun_out/r04d/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], "| w16 %.3f ms" % d["long_piece_model"]["ms_per_step"], d["long_piece_model"]["kernels_ms"])
    except Exception as e:
        print(v, "failed", e)
PY
( time timeout 900 python -m pytest tests/test_full_size.py -m gpu -x -q -k "full_size_sample or open_vocab" ) > $O/fullsize.log 2>&1; tail -4 $O/fullsize.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
