# usage (GPU box): bash scripts/r04_exp1.sh  -- round 4, first measurement session: the word-form GPU tests, then the
# word kernel's experiment builds (wave.h SPMX_EXP 8 / 16 / 32: one more text gather, no id bursts, one more memo
# gather per iteration) on the C2 bench line without side configs.  Results under gpurun_out/r04b/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; mkdir -p $O
( time timeout 900 python -m pytest tests/test_word_form.py -m gpu -x -q --durations=8 ) > $O/wordform.log 2>&1; tail -5 $O/wordform.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-model --no-side-configs"
timeout 200 $B > $O/bench_base.json 2> $O/bench_base.err
for v in 8 16 32; do
  SPMX_LIB=$PWD/sentencepiece_amd/variants/libspmx_exp$v.so timeout 200 $B > $O/bench_exp$v.json 2> $O/bench_exp$v.err
done
python - <<'PY'
import json
for v in ("base", "exp8", "exp16", "exp32"):
    try:
        d = json.load(open("gpurun_out/r04b/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"])
    except Exception as e:
        print(v, "failed", e)
PY
