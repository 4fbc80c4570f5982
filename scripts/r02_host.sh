# usage (GPU box): bash scripts/r02_host.sh  -- host-buffer form: SDMA engines vs blit kernels for the PCIe copies
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { echo "$1 threads=$2 chunk=$3: $(env $1 HOST_RATE_ONLY=flat SPMX_HOST_THREADS=$2 SPMX_HOST_CHUNK=$3 timeout 200 python scripts/host_rate.py 10000000 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["flat"]["sentences_per_s"]/1e6,1), "M/s", round(d["flat"]["ms"],1), "ms")')"; }
run HSA_ENABLE_SDMA=1 24 0
run HSA_ENABLE_SDMA=0 24 0
run HSA_ENABLE_SDMA=0 24 500000
run HSA_ENABLE_SDMA=0 12 0
run HSA_ENABLE_SDMA=0 32 250000
run HSA_ENABLE_SDMA=1 24 0
