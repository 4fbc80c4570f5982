# usage (GPU box): bash scripts/r04_exp11.sh -- workgroup widths re-tuned on the window-reuse word loop.  gpurun_out/r04m/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04m; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --no-second-model"
run() { name=$1; shift; env "$@" timeout 200 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run head SPMX_X=0
run ww14 SPMX_WORD_WAVES=14
run ww16 SPMX_WORD_WAVES=16
run ww10 SPMX_WORD_WAVES=10
run fw5 SPMX_FORK_WAVES=5
run fw3 SPMX_FORK_WAVES=3
run ww14fw3 SPMX_WORD_WAVES=14 SPMX_FORK_WAVES=3
run noov SPMX_NO_OVERLAP=1
python - <<'PY'
import json
for v in ("head", "ww14", "ww16", "ww10", "fw5", "fw3", "ww14fw3", "noov"):
    try:
        d = json.load(open("gpurun_out/r04m/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"])
    except Exception as e:
        print(v, "failed", e)
PY
