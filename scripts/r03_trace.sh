# usage (GPU box): bash scripts/r03_trace.sh <tag> [bench args]  -- rocprofv3 --kernel-trace --stats of the bench command, summary kept
TAG=${1:-r03}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs "$@" > $O/bench.json 2> $O/bench.err
DB=$(find $O/prof -name "x_results.db" | head -1)
python scripts/rocpd_summary.py "$DB" > $O/kernel_stats.txt 2>&1
cat $O/kernel_stats.txt | cut -c1-150
tail -c 1200 $O/bench.json
rm -rf $O/prof
