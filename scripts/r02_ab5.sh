# usage (GPU box): bash scripts/r02_ab5.sh <tag>  -- headline after the normalizer change; sub-bucket counts on unsorted input; host form sweep
TAG=${1:-r02n}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
BENCH_ARGS="--no-second-model" bash scripts/r02_ab.sh $TAG "SPMX_X=0"
BENCH_ARGS="--unsorted --no-second-model" bash scripts/r02_ab.sh ${TAG}_unsorted "SPMX_X=0" "SPMX_SUB_BUCKETS=32" "SPMX_SUB_BUCKETS=64"
BENCH_ARGS="--model bpe32k --no-second-model" bash scripts/r02_ab.sh ${TAG}_bpe "SPMX_X=0"
for cfg in "24 0" "24 500000" "32 500000" "32 1000000" "16 1000000" "48 250000"; do
  set -- $cfg
  echo "host threads=$1 chunk=$2: $(HOST_RATE_ONLY=flat SPMX_HOST_THREADS=$1 SPMX_HOST_CHUNK=$2 timeout 200 python scripts/host_rate.py 10000000 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["flat"]["sentences_per_s"]/1e6,1), "M/s", round(d["flat"]["ms"],1), "ms")')"
done
