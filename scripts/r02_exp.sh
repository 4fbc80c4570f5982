# usage (GPU box): bash scripts/r02_exp.sh <tag> <variant> ...  -- WRITE_SIZE (and FETCH_SIZE) of the experiment builds
TAG=${1:-r02x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/sentencepiece_amd/variants
for v in "$@"; do
  echo "==== $v"
  SPMX_LIB=$V/libspmx_$v.so COUNTERS="${COUNTERS:-WRITE_SIZE}" PASS_TIMEOUT=120 timeout 300 bash scripts/pmc_traffic.sh ${TAG}_$v 10000000 uni32k 2>&1 | grep -E "Encode|rc="
done
