# usage (GPU box): bash scripts/r02_last.sh <tag>  -- the head's default bench line (traffic from the committed PMC record) and its rocprofv3 kernel stats
TAG=${1:-r02last}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
timeout 300 python bench.py > $O/bench_uni32k_10m.json 2> $O/bench.err; tail -c 400 $O/bench_uni32k_10m.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o uni -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/uni_results.db $O/prof/uni_results.db 2>/dev/null | head -1)
python scripts/rocpd_summary.py "$DB" > $O/uni32k_10m_kernel_stats.txt 2>> $O/prof.err; head -6 $O/uni32k_10m_kernel_stats.txt
rm -rf $O/prof
