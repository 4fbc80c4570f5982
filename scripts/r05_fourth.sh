TAG=${1:-r05d}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "%.4g" % d["value"], "%.3f ms" % d["ms_per_step"], json.dumps(r.get("all_kernels_ms")), "phase", json.dumps(r.get("phase_cycles")))
PY
}
SPMX_NO_OVERLAP=1 timeout 600 python bench.py --no-cpu-baseline --no-side-configs --no-second-model --steps 4 --warmup 2 > $O/bench_alone.json 2> $O/bench_alone.err; show $O/bench_alone.json
