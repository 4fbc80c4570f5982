# usage (GPU box): bash scripts/r02_round.sh <tag>   -- the round's evidence: parity suite, headline bench line (with the CPU
# baseline), rocprofv3 kernel stats of the same command, PMC traffic passes, side benches
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench_uni32k_10m.json 2> $O/bench_uni32k_10m.err; tail -c 3000 $O/bench_uni32k_10m.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o uni -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/uni_results.db $O/prof/uni_results.db 2>/dev/null | head -1)
python scripts/rocpd_summary.py "$DB" > $O/uni32k_10m_kernel_stats.txt 2>> $O/prof.err; head -14 $O/uni32k_10m_kernel_stats.txt
rm -rf $O/prof
timeout 900 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k 2>&1 | tail -12
timeout 600 python bench.py --model bpe32k --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_bpe32k_10m.json 2> $O/bench_bpe.err; tail -c 1200 $O/bench_bpe32k_10m.json
timeout 900 python bench.py --model c5_250k --sentences 1000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5_250k_1m.json 2> $O/bench_c5.err; tail -c 1200 $O/bench_c5_250k_1m.json
