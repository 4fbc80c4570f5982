# usage (GPU box): bash scripts/r02_ab2.sh <tag>  -- short back-pointer form A/B, PMC traffic, new GPU tests, lattice rates
TAG=${1:-r02j}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_sampling.py tests/test_decode.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_new.txt
BENCH_ARGS="--no-second-model" bash scripts/r02_ab.sh $TAG "SPMX_X=0" "SPMX_NO_BP_SHORT=1" "SPMX_TILE_WAVES=14" "SPMX_TILE_WAVES=12"
BENCH_ARGS="--model bpe1k_llama --sentences 4000000" bash scripts/r02_ab.sh ${TAG}_llama "SPMX_X=0" "SPMX_WORDTAB_MIN=1"
date; PASS_TIMEOUT=150 timeout 400 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k 2>&1 | tail -14; date
timeout 300 python scripts/lattice_rate.py 200000 2>&1 | tail -2 | tee $O/lattice_rate.json
