# usage (GPU box): bash scripts/r04_exp4.sh -- round 4, fourth session: two-piece 16-byte memo entries, the quad compact
# kernel, the general launch sized to one tile per wavefront; the gather over the real RCCL.  Results: gpurun_out/r04e/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04e; mkdir -p $O
( time timeout 600 python -m pytest tests/test_word_form.py tests/test_gather.py tests/test_gpu_parity.py -m gpu -x -q --durations=5 ) > $O/tests.log 2>&1; tail -4 $O/tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs"
timeout 200 $B > $O/bench_auto.json 2> $O/bench_auto.err
for v in 4 6 8; do
  SPMX_FORK_WAVES=$v timeout 200 $B > $O/bench_fw$v.json 2> $O/bench_fw$v.err
done
SPMX_NO_OVERLAP=1 timeout 200 $B > $O/bench_noov.json 2> $O/bench_noov.err
python - <<'PY'
import json
for v in ("auto", "fw4", "fw6", "fw8", "noov"):
    try:
        d = json.load(open("gpurun_out/r04e/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], "| w16 %.3f ms" % d["long_piece_model"]["ms_per_step"], d["long_piece_model"]["kernels_ms"])
    except Exception as e:
        print(v, "failed", e)
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --no-second-model > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err; cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head -3
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200 > $O/kernel_stats_head.txt; cat $O/kernel_stats_head.txt 2>/dev/null | cut -c1-160
rm -rf $O/prof
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
