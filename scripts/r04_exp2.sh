# usage (GPU box): bash scripts/r04_exp2.sh  -- round 4, second session: the plain scan + early general launch.
# GPU tests of the word form and the every-sentence full-size checks, then the C2 line for a few wavefront caps of the
# general launch (SPMX_FORK_WAVES) and without the scan, then the default bench line.  Results under gpurun_out/r04c/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c; mkdir -p $O
( time timeout 600 python -m pytest tests/test_word_form.py -m gpu -x -q --durations=5 ) > $O/wordform.log 2>&1; tail -4 $O/wordform.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs"
for v in 4 8 2 16; do
  SPMX_FORK_WAVES=$v timeout 200 $B > $O/bench_fw$v.json 2> $O/bench_fw$v.err
done
SPMX_NO_SCAN=1 timeout 200 $B > $O/bench_noscan.json 2> $O/bench_noscan.err
python - <<'PY'
import json
for v in ("fw4", "fw8", "fw2", "fw16", "noscan"):
    try:
        d = json.load(open("gpurun_out/r04c/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], "| w16 %.3f ms" % d["long_piece_model"]["ms_per_step"], d["long_piece_model"]["kernels_ms"])
    except Exception as e:
        print(v, "failed", e)
PY
( time timeout 900 python -m pytest tests/test_full_size.py -m gpu -x -q --durations=10 ) > $O/fullsize.log 2>&1; tail -14 $O/fullsize.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
