# usage (GPU box): bash scripts/run_final.sh <tag>
# The round's evidence in one call: HBM traffic (two --pmc passes) -> profiles/pmc_traffic.json, GPU parity tests,
# the headline bench line (which reads that traffic), rocprofv3 kernel stats of the same command.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
bash scripts/pmc_traffic.sh $TAG > $O/pmc_traffic.log 2>&1; tail -12 $O/pmc_traffic.log
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/pmc_traffic.json
cd /tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_uni32k_10m.json 2> $O/bench_uni32k_10m.err; tail -c 2600 $O/bench_uni32k_10m.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o uni -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/uni_results.db $O/prof/uni_results.db 2>/dev/null | head -1)
python scripts/rocpd_summary.py "$DB" > $O/uni32k_10m_kernel_stats.txt 2>> $O/prof.err; head -14 $O/uni32k_10m_kernel_stats.txt
rm -rf $O/prof
# A/B of the tile queue (SPMX_TILE_ORDER=asc: shortest tiles first; SPMX_STATIC_TILES=1: fixed stride)
SPMX_TILE_ORDER=asc timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_tiles_asc.json 2>> $O/prof.err
python - $O <<'PY'
import json, sys
for f in ("bench_uni32k_10m", "bench_tiles_asc"):
    d = json.load(open(sys.argv[1] + "/" + f + ".json")); print(f, round(d["value"] / 1e6, 1), round(d["ms_per_step"], 2), d["roofline"]["all_kernels_ms"])
PY
