# usage (GPU box): bash scripts/r02_ab4.sh <tag>  -- headline, PMC traffic, side configs after a kernel change
TAG=${1:-r02l}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
BENCH_ARGS="--no-second-model" bash scripts/r02_ab.sh $TAG "SPMX_X=0" "SPMX_NO_BP_SHORT=1"
date; PASS_TIMEOUT=150 timeout 400 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k 2>&1 | tail -12; date
BENCH_ARGS="--model uni32k_w16 --no-second-model" bash scripts/r02_ab.sh ${TAG}_w16 "SPMX_X=0"
BENCH_ARGS="--unsorted --no-second-model" bash scripts/r02_ab.sh ${TAG}_unsorted "SPMX_X=0"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
