# usage (GPU box): bash scripts/r03_final2.sh <tag>  -- after the lattice piece forms (kernels_nbest.h changed, so the PMC
# records are re-made for the new source hash): the GPU tests of what changed, the PMC traffic passes, the default
# bench line, the lattice rates.
TAG=${1:-r03f}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 400 python -m pytest tests/test_lattice_pieces.py tests/test_nbest.py tests/test_sampling.py tests/test_cpp_facade.py tests/test_capi.py tests/test_gpu_parity.py -m gpu -x -q -k "not no_length_limit and not document" ) 2>&1 | tail -6 | tee $O/pytest_gpu_changed.txt
PASS_TIMEOUT=100 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k > $O/pmc_traffic_uni.log 2>&1; tail -3 $O/pmc_traffic_uni.log | cut -c1-200
PASS_TIMEOUT=100 bash scripts/pmc_traffic.sh $TAG 1000000 c5_250k > $O/pmc_traffic_c5.log 2>&1
PASS_TIMEOUT=100 bash scripts/pmc_traffic.sh $TAG 10000000 bpe32k > $O/pmc_traffic_bpe.log 2>&1
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
out = {}
note = ""
for m in ("uni32k", "c5_250k", "bpe32k"):
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
timeout 500 python bench.py > $O/bench_uni32k_10m.json 2> $O/bench.err; tail -c 300 $O/bench_uni32k_10m.json
timeout 150 python scripts/lattice_rate.py 2>/dev/null | tail -1 | tee $O/lattice_rate.json
ls $O
