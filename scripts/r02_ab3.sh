# usage (GPU box): bash scripts/r02_ab3.sh <tag>  -- position-major back-pointer blocks: headline + PMC traffic; non-temporal variants
TAG=${1:-r02k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
V=$GRAFT_REPO_ROOT/sentencepiece_amd/variants
BENCH_ARGS="--no-second-model" bash scripts/r02_ab.sh $TAG "SPMX_X=0" "SPMX_LIB=$V/libspmx_nt1.so" "SPMX_LIB=$V/libspmx_nt5.so" "SPMX_LIB=$V/libspmx_nt7.so" "SPMX_LIB=$V/libspmx_nt15.so" "SPMX_NO_BP_SHORT=1"
date; PASS_TIMEOUT=150 timeout 400 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k 2>&1 | tail -8; date
BENCH_ARGS="--model c5_250k --sentences 1000000" bash scripts/r02_ab.sh ${TAG}_c5 "SPMX_X=0"
BENCH_ARGS="--model uni32k_w16 --no-second-model" bash scripts/r02_ab.sh ${TAG}_w16 "SPMX_X=0"
