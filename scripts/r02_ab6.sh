# usage (GPU box): bash scripts/r02_ab6.sh <tag>  -- backtrack prefetch variants; PMC traffic of the BPE kernel
TAG=${1:-r02q}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/sentencepiece_amd/variants
BENCH_ARGS="--no-second-model" bash scripts/r02_ab.sh $TAG "SPMX_X=0" "SPMX_LIB=$V/libspmx_pf1.so" "SPMX_LIB=$V/libspmx_pf2.so" "SPMX_LIB=$V/libspmx_pf3.so" "SPMX_X=1"
PASS_TIMEOUT=150 timeout 400 bash scripts/pmc_traffic.sh ${TAG}_bpe 10000000 bpe32k 2>&1 | tail -12
