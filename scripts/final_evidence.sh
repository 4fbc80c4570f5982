# usage (GPU box): bash scripts/final_evidence.sh <tag>  -- the round's evidence on the head's sources, most important first:
# smoke, PMC traffic passes (every record of the bench line) merged into profiles/pmc_traffic.json, rocprofv3 kernel stats
# (uni32k, c5_250k), SQ counters, then the default bench line (which picks the fresh traffic records up).  SKIP_TESTS=1
# leaves the full GPU suite out (run it separately: python -m pytest tests -m gpu).  QUICK=1: headline + C5 only.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then ( time timeout 1200 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt; fi
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k > $O/pmc_traffic_uni.log 2>&1; tail -3 $O/pmc_traffic_uni.log | cut -c1-200
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 1000000 c5_250k > $O/pmc_traffic_c5.log 2>&1
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 1000000 c5_250k_bf > $O/pmc_traffic_c5bf.log 2>&1
if [ -z "$QUICK" ]; then
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 10000000 bpe32k > $O/pmc_traffic_bpe.log 2>&1
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k_w16 > $O/pmc_traffic_w16.log 2>&1
PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 10000000 bpe1k_llama > $O/pmc_traffic_llama.log 2>&1
CORPUS=open_vocab PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k > $O/pmc_traffic_ov.log 2>&1
CORPUS=botchan PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 8576000 uni32k > $O/pmc_traffic_botchan.log 2>&1
CORPUS=docs_16k PASS_TIMEOUT=120 bash scripts/pmc_traffic.sh $TAG 8192 uni32k > $O/pmc_traffic_docs16k.log 2>&1
CORPUS=docs_1m PASS_TIMEOUT=150 bash scripts/pmc_traffic.sh $TAG 256 uni32k > $O/pmc_traffic_docs1m.log 2>&1
fi
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
out = {}
note = ""
for m in ("uni32k", "c5_250k", "bpe32k", "c5_250k_bf", "uni32k_w16", "bpe1k_llama", "uni32k@open_vocab", "uni32k@botchan", "uni32k@docs_16k", "uni32k@docs_1m"):
    try:
        d = json.load(open("%s/pmc_traffic_%s.json" % (O, m)))
    except Exception as e:
        print("missing", m, e); continue
    note = d.pop("_note", note)
    out.update(d)
out["_note"] = note + "; made by scripts/pmc_traffic.sh (bench.py --no-side-configs, steps 2, warmup 1) on the kernel sources whose hash each record carries"
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(out, open(O + "/pmc_traffic.json", "w"), indent=1)
PY
for M in uni32k c5_250k; do
  NS=10000000; [ $M = c5_250k ] && NS=1000000
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --model $M --sentences $NS --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs > $O/trace_bench_$M.json 2> $O/trace_$M.err
  DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/${M}_kernel_stats.txt 2>&1; rm -rf $O/prof
done
head -8 $O/uni32k_kernel_stats.txt | cut -c1-150
if [ -z "$QUICK" ]; then
GROUPS_MAX=2 bash scripts/pmc_sq.sh 2000000 > $O/uni32k_2m_pmc_sq.txt 2>&1; rm -rf gpurun_out/pmc_sq
MODEL=c5_250k GROUPS_MAX=2 bash scripts/pmc_sq.sh 1000000 > $O/c5_1m_pmc_sq.txt 2>&1; rm -rf gpurun_out/pmc_sq
fi
cp profiles/pmc_traffic.json $O/pmc_traffic_used.json
( time timeout 1500 python bench.py > $O/bench_uni32k_10m.json 2> $O/bench.err ) 2> $O/bench_wall.txt; tail -3 $O/bench_wall.txt; tail -c 600 $O/bench_uni32k_10m.json
ls $O
