# usage (GPU box): bash scripts/r02_round.sh <tag>   -- the round's evidence: parity suite, PMC traffic passes (installed
# into profiles/pmc_traffic.json so that the bench lines that follow carry `traffic`), headline bench line (with the CPU
# baseline), rocprofv3 kernel stats of the same command, side benches, lattice and host-form rates
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
PASS_TIMEOUT=150 timeout 400 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k 2>&1 | tail -6
python - $O <<'PY'
import json, sys, os
o = sys.argv[1]
new = json.load(open(os.path.join(o, "pmc_traffic_uni32k.json")))
path = "profiles/pmc_traffic.json"
cur = {}
if os.path.exists(path):
    cur = {k: v for k, v in json.load(open(path)).items() if isinstance(v, dict) or k == "_note"}
cur.update(new)
json.dump(cur, open(path, "w"), indent=1)
json.dump(cur, open(os.path.join(o, "pmc_traffic.json"), "w"), indent=1)
PY
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench_uni32k_10m.json 2> $O/bench_uni32k_10m.err; tail -c 3500 $O/bench_uni32k_10m.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o uni -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/uni_results.db $O/prof/uni_results.db 2>/dev/null | head -1)
python scripts/rocpd_summary.py "$DB" > $O/uni32k_10m_kernel_stats.txt 2>> $O/prof.err; head -14 $O/uni32k_10m_kernel_stats.txt
rm -rf $O/prof
timeout 600 python bench.py --model bpe32k --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_bpe32k_10m.json 2> $O/bench_bpe.err; tail -c 600 $O/bench_bpe32k_10m.json
timeout 900 python bench.py --model c5_250k --sentences 1000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5_250k_1m.json 2> $O/bench_c5.err; tail -c 600 $O/bench_c5_250k_1m.json
timeout 600 python bench.py --unsorted --steps 5 --warmup 2 --no-cpu-baseline --no-second-model > $O/bench_uni32k_10m_unsorted.json 2> $O/bench_unsorted.err; tail -c 600 $O/bench_uni32k_10m_unsorted.json
timeout 600 python bench.py --model bpe1k_llama --sentences 4000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_bpe1k_llama_4m.json 2> $O/bench_llama.err; tail -c 400 $O/bench_bpe1k_llama_4m.json
timeout 300 python scripts/lattice_rate.py 200000 2>/dev/null | tail -1 | tee $O/lattice_rate.json
timeout 600 python scripts/host_rate.py 10000000 2>/dev/null | tail -1 | tee $O/host_rate.json
