# usage (GPU box): bash scripts/r04_exp6.sh -- round 4: issue priority of the word kernels next to the general launch,
# the reverted compact kernel, the wider plain scan; kernel stats of the head.  Results: gpurun_out/r04g/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04g; mkdir -p $O
( time timeout 600 python -m pytest tests/test_word_form.py -m gpu -x -q ) > $O/tests.log 2>&1; tail -3 $O/tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs"
run() { name=$1; shift; env "$@" timeout 200 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run head SPMX_X=0
run noprio SPMX_NO_WORD_PRIO=1
run fw4 SPMX_FORK_WAVES=4
run fw4_noprio SPMX_FORK_WAVES=4 SPMX_NO_WORD_PRIO=1
run fw12 SPMX_FORK_WAVES=12
run head2 SPMX_X=0
python - <<'PY'
import json
for v in ("head", "noprio", "fw4", "fw4_noprio", "fw12", "head2"):
    try:
        d = json.load(open("gpurun_out/r04g/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], "| w16 %.3f ms" % d["long_piece_model"]["ms_per_step"], d["long_piece_model"]["kernels_ms"])
    except Exception as e:
        print(v, "failed", e)
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- $B --no-second-model > /dev/null 2> $O/prof.err
DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/kernel_stats_head.txt 2>&1; rm -rf $O/prof; head -14 $O/kernel_stats_head.txt | cut -c1-150
