# usage (GPU box): bash scripts/r02_ab7.sh <tag>  -- BPE after the grouped id stores: bench, PMC traffic; headline + its PMC passes again (kernel sources changed)
TAG=${1:-r02r}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
BENCH_ARGS="--model bpe32k --no-second-model" bash scripts/r02_ab.sh ${TAG}_bpe "SPMX_X=0"
BENCH_ARGS="--model bpe1k_llama --sentences 4000000" bash scripts/r02_ab.sh ${TAG}_llama "SPMX_X=0"
PASS_TIMEOUT=150 timeout 400 bash scripts/pmc_traffic.sh ${TAG} 10000000 bpe32k 2>&1 | grep -E "fetch_kb|write_kb|bytes"
PASS_TIMEOUT=150 timeout 400 bash scripts/pmc_traffic.sh ${TAG} 10000000 uni32k 2>&1 | grep -E "fetch_kb|write_kb|bytes"
BENCH_ARGS="--no-second-model" bash scripts/r02_ab.sh $TAG "SPMX_X=0"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bpe or golden" 2>&1 | tail -2
