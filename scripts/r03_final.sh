# usage (GPU box): bash scripts/r03_final.sh <tag>  -- the committed head once more, most important first: full GPU suite,
# smoke(), the default bench line, the PMC traffic passes (uni32k, c5_250k, bpe32k), the rocprofv3 kernel trace, the SQ
# pass, the host forms.  Summaries land in gpurun_out/<tag>/ (copied to profiles/ by hand).
TAG=${1:-r03z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=25 ) 2>&1 | tail -40 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 600 python bench.py > $O/bench_uni32k_10m.json 2> $O/bench.err; tail -c 300 $O/bench_uni32k_10m.json
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k > $O/pmc_traffic_uni.log 2>&1; tail -5 $O/pmc_traffic_uni.log
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 1000000 c5_250k > $O/pmc_traffic_c5.log 2>&1
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 10000000 bpe32k > $O/pmc_traffic_bpe.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs > $O/trace_bench.json 2> $O/trace.err
DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/uni32k_10m_kernel_stats.txt 2>&1; rm -rf $O/prof
timeout 300 python scripts/host_rate.py 4000000 > $O/host_rate_4m.json 2> $O/host_rate.err; tail -c 600 $O/host_rate_4m.json
timeout 200 python bench.py --model c5_250k --sentences 1000000 --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs 2>/dev/null | tail -1 > $O/bench_c5_250k_1m.json
GROUPS_MAX=2 bash scripts/pmc_sq.sh 2000000 > $O/uni32k_2m_pmc_sq.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --model bpe32k --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs > $O/trace_bench_bpe.json 2> $O/trace_bpe.err
DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/bpe32k_10m_kernel_stats.txt 2>&1; rm -rf $O/prof
ls -la $O
