# the general launch beside the word-per-lane rounds: wavefront counts / CU partitions (C2, uni32k)
TAG=${1:-r05h}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[2], "%.4g" % d["value"], "%.3f ms" % d["ms_per_step"], json.dumps(r.get("all_kernels_ms")))
PY
}
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-side-configs --no-second-model --steps 4 --warmup 2 > $O/b.json 2> $O/b.err; show $O/b.json "$*"; }
run SPMX_WORDWAVE_WAVES=12 SPMX_FORK_WAVES=4
run SPMX_WORDWAVE_WAVES=9 SPMX_FORK_WAVES=3
run SPMX_WORDWAVE_WAVES=10 SPMX_FORK_WAVES=2
run SPMX_WORDWAVE_WAVES=12 SPMX_FORK_WAVES=16 SPMX_FORK_CUS=32
run SPMX_WORDWAVE_WAVES=12 SPMX_FORK_WAVES=16 SPMX_FORK_CUS=48
run SPMX_WORDWAVE_WAVES=12 SPMX_FORK_WAVES=16 SPMX_FORK_CUS=64
run SPMX_WORDWAVE_WAVES=12 SPMX_FORK_WAVES=8 SPMX_FORK_CUS=96
