# usage (GPU box): bash scripts/r05_end.sh <tag>  -- the round's last GPU call, after the collecting round's tag loads became
# plain loads (kernel sources changed: the PMC passes behind `traffic` are taken again): the default bench line first (every
# record's ids against the compiled reference over the whole batch), then the PMC passes most important first, then kernel stats
TAG=${1:-r05end}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 400 python bench.py > $O/bench_uni32k_10m.json 2> $O/bench.err ) 2> $O/bench_wall.txt; tail -c 300 $O/bench_uni32k_10m.json; echo
P() { CORPUS=${4:-synthetic} PASS_TIMEOUT=60 bash scripts/pmc_traffic.sh $TAG $2 $1 > $O/pmc_traffic_$3.log 2>&1; echo "pmc $3 done"; }
P uni32k 10000000 uni
P bpe1k_llama 10000000 llama
P bpe32k 10000000 bpe
P c5_250k 1000000 c5
P uni32k 10000000 ov open_vocab
P uni32k 8576000 botchan botchan
P uni32k_w16 10000000 w16
P c5_250k_bf 1000000 c5bf
P uni32k 8192 docs16k docs_16k
P uni32k 256 docs1m docs_1m
timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs > $O/trace_bench_uni32k.json 2> $O/trace_uni32k.err
DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/uni32k_10m_kernel_stats.txt 2>&1; rm -rf $O/prof
head -8 $O/uni32k_10m_kernel_stats.txt | cut -c1-150
