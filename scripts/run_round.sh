# usage (GPU box): bash scripts/run_round.sh <tag>   -- parity tests, headline bench, rocprofv3 kernel stats, side benches
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench_uni32k_10m.json 2> $O/bench_uni32k_10m.err; tail -c 2500 $O/bench_uni32k_10m.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o uni -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/uni_results.db $O/prof/uni_results.db 2>/dev/null | head -1)
python scripts/rocpd_summary.py "$DB" > $O/uni32k_10m_kernel_stats.txt 2>> $O/prof.err; head -12 $O/uni32k_10m_kernel_stats.txt
rm -rf $O/prof


timeout 600 python bench.py --model bpe32k --steps 3 --warmup 1 > $O/bench_bpe32k_10m.json 2> $O/bench_bpe.err; tail -c 1500 $O/bench_bpe32k_10m.json
timeout 900 python bench.py --model c5_250k --sentences 1000000 --steps 3 --warmup 1 > $O/bench_c5_250k_1m.json 2> $O/bench_c5.err; tail -c 1500 $O/bench_c5_250k_1m.json

# the rows next to the hot path: decode, spans / normalize, line splitter, n-best
timeout 300 python scripts/decode_rate.py > $O/decode_rate.json 2>> $O/side.err; tail -c 600 $O/decode_rate.json
timeout 300 python scripts/spans_rate.py > $O/spans_rate.json 2>> $O/side.err; tail -c 600 $O/spans_rate.json
timeout 300 python scripts/split_rate.py > $O/split_rate.json 2>> $O/side.err; tail -c 600 $O/split_rate.json
timeout 300 python scripts/nbest_rate.py > $O/nbest_rate.json 2>> $O/side.err; tail -c 600 $O/nbest_rate.json
