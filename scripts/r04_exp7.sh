# usage (GPU box): bash scripts/r04_exp7.sh -- round 4: the linear plain scan; fork widths 4 / 5 / 6; kernel stats.  gpurun_out/r04i/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04i; mkdir -p $O
( time timeout 600 python -m pytest tests/test_word_form.py -m gpu -x -q ) > $O/tests.log 2>&1; tail -3 $O/tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs"
run() { name=$1; shift; env "$@" timeout 200 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run fw4 SPMX_X=0
run w16probe SPMX_X=1


run fw4b SPMX_X=0
python - <<'PY'
import json
for v in ("fw4", "w16probe", "fw4b"):
    try:
        d = json.load(open("gpurun_out/r04i/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], "| w16 %.3f ms" % d["long_piece_model"]["ms_per_step"], d["long_piece_model"]["kernels_ms"])
    except Exception as e:
        print(v, "failed", e)
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- $B --no-second-model > /dev/null 2> $O/prof.err
DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/kernel_stats_head.txt 2>&1; rm -rf $O/prof; head -15 $O/kernel_stats_head.txt | cut -c1-150
