# usage (GPU box): bash scripts/r04_exp8.sh -- the wide plain-scan launch, kernel stats.  gpurun_out/r04j/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04j; mkdir -p $O
( time timeout 600 python -m pytest tests/test_word_form.py -m gpu -x -q -k "plain_scan or nul" ) > $O/tests.log 2>&1; tail -3 $O/tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs"
timeout 200 $B > $O/bench_head.json 2> $O/bench_head.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04j/bench_head.json"))
print("head %.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], "| w16 %.3f ms" % d["long_piece_model"]["ms_per_step"], d["long_piece_model"]["kernels_ms"])
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- $B --no-second-model > /dev/null 2> $O/prof.err
DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/kernel_stats_head.txt 2>&1; rm -rf $O/prof; head -15 $O/kernel_stats_head.txt | cut -c1-150
