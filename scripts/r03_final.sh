# usage (GPU box): bash scripts/r03_final.sh <tag>  -- the committed head once more: full GPU suite, smoke(), the default bench
# line, its rocprofv3 kernel trace, the PMC traffic / SQ passes, the rates of the neighbouring forms.  Summaries land in
# gpurun_out/<tag>/ (copied to profiles/ by hand).
TAG=${1:-r03z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_uni32k_10m.json 2> $O/bench.err; tail -c 400 $O/bench_uni32k_10m.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs > $O/trace_bench.json 2> $O/trace.err
DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/uni32k_10m_kernel_stats.txt 2>&1; rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --model bpe32k --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs > $O/trace_bench_bpe.json 2> $O/trace_bpe.err
DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/bpe32k_10m_kernel_stats.txt 2>&1; rm -rf $O/prof
PASS_TIMEOUT=150 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k > $O/pmc_traffic_uni.log 2>&1
PASS_TIMEOUT=150 bash scripts/pmc_traffic.sh $TAG 10000000 bpe32k > $O/pmc_traffic_bpe.log 2>&1
bash scripts/pmc_sq.sh 2000000 > $O/uni32k_2m_pmc_sq.txt 2>&1
timeout 600 python scripts/host_rate.py 10000000 > $O/host_rate.json 2> $O/host_rate.err; tail -c 900 $O/host_rate.json
timeout 300 python scripts/docs_rate.py --docs 8192 --bytes 16384 2>/dev/null | tail -1 > $O/docs_16k_uni32k.json
timeout 300 python scripts/docs_rate.py --docs 256 --bytes 1048576 2>/dev/null | tail -1 > $O/docs_1m_uni32k.json
timeout 300 python scripts/docs_rate.py --model bpe32k --docs 8192 --bytes 16384 --cpu-seconds 1 2>/dev/null | tail -1 > $O/docs_16k_bpe32k.json
timeout 300 python bench.py --unsorted --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs 2>/dev/null | tail -1 > $O/bench_uni32k_10m_unsorted.json
timeout 300 python bench.py --model bpe1k_llama --sentences 4000000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bpe1k_llama_4m.json
timeout 300 python scripts/lattice_rate.py 2>/dev/null | tail -1 > $O/lattice_rate.json
ls -la $O
