# usage (GPU box): bash scripts/r02_cb.sh <tag>  -- child-label hash c & 31 against (c * 37 >> 3) & 31; PMC passes of the new build
TAG=${1:-r02y}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/sentencepiece_amd/variants
BENCH_ARGS="--no-second-model" bash scripts/r02_ab.sh $TAG "SPMX_X=0" "SPMX_LIB=$V/libspmx_base.so" "SPMX_X=1"
PASS_TIMEOUT=100 timeout 260 bash scripts/pmc_traffic.sh $TAG 10000000 uni32k 2>&1 | grep -E "fetch_kb|write_kb|bytes|src_sha"
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden" 2>&1 | tail -1
