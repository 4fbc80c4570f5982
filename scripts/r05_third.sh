# usage (GPU box): bash scripts/r05_third.sh <tag> -- the word-per-lane first round WITHOUT the general launch beside it (SPMX_NO_OVERLAP=1), by wavefronts per workgroup
TAG=${1:-r05c}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "%.4g" % d["value"], "%.3f ms" % d["ms_per_step"], json.dumps(r.get("all_kernels_ms")), "trips", (r.get("phase_cycles") or {}).get("search_trips"), "cyc", (r.get("phase_cycles") or {}).get("segment"))
PY
}
for WV in 12 8 6 4; do
  SPMX_NO_OVERLAP=1 SPMX_WORDWAVE_WAVES=$WV timeout 600 python bench.py --no-cpu-baseline --no-side-configs --no-second-model --steps 4 --warmup 2 > $O/bench_alone_wv$WV.json 2> $O/bench_alone_wv$WV.err; show $O/bench_alone_wv$WV.json
done
SPMX_NO_OVERLAP=1 SPMX_WORD_WAVE=0 timeout 600 python bench.py --no-cpu-baseline --no-side-configs --no-second-model --steps 4 --warmup 2 > $O/bench_alone_old.json 2> $O/bench_alone_old.err; show $O/bench_alone_old.json
