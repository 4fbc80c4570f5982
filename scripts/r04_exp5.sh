# usage (GPU box): bash scripts/r04_exp5.sh -- round 4, A/B of the last three changes one by one (two-piece 16-byte memo
# entries, 16-bit ids in the word kernels' arena slots, quad compact kernel), each with rocprofv3 kernel stats.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --no-second-model"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 $B > $O/bench_$name.json 2> $O/bench_$name.err
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o x -- $B > /dev/null 2> $O/prof_$name.err
  DB=$(find $O/prof_$name -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/kernel_stats_$name.txt 2>&1; rm -rf $O/prof_$name
}
run head SPMX_X=0
run ids32 SPMX_NO_IDS16=1
run memo_one SPMX_MEMO16_ONE=1
run memo_one_ids32 SPMX_MEMO16_ONE=1 SPMX_NO_IDS16=1
python - <<'PY'
import json
for v in ("head", "ids32", "memo_one", "memo_one_ids32"):
    try:
        d = json.load(open("gpurun_out/r04f/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"])
    except Exception as e:
        print(v, "failed", e)
    try:
        for l in open("gpurun_out/r04f/kernel_stats_%s.txt" % v).read().split("\n")[1:12]:
            print("   ", l[:130])
    except Exception as e:
        print("  no stats", e)
PY
