# usage (GPU box): bash scripts/r04_exp9.sh -- issue priority for the GENERAL launch next to the first word round.  gpurun_out/r04k/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04k; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs"
run() { name=$1; shift; env "$@" timeout 200 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run prio SPMX_X=0
run noprio SPMX_NO_FORK_PRIO=1
run prio_fw3 SPMX_FORK_WAVES=3
run prio_fw2 SPMX_FORK_WAVES=2
run prio_fw6 SPMX_FORK_WAVES=6
run prio2 SPMX_X=0
python - <<'PY'
import json
for v in ("prio", "noprio", "prio_fw3", "prio_fw2", "prio_fw6", "prio2"):
    try:
        d = json.load(open("gpurun_out/r04k/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], "| w16 %.3f ms" % d["long_piece_model"]["ms_per_step"], d["long_piece_model"]["kernels_ms"])
    except Exception as e:
        print(v, "failed", e)
PY
