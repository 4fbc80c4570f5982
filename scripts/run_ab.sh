# usage: bash scripts/run_ab.sh [sentences]  -- A/B of the unigram kernel variants on one box (no CPU baseline)
N=${1:-4000000}
run() {
  env "$@" timeout 600 python bench.py --sentences $N --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; pc=r['phase_cycles']
    print('$*', 'Msent/s %.1f'%(d['value']/1e6), r['all_kernels_ms'], pc, 'cyc/trip %.0f'%(pc['segment']/max(pc['search_trips'],1)))
except Exception as e:
    print('$*', 'FAILED', e)"
}
run SPMX_AB=default
run SPMX_TILE_WAVES=8
run SPMX_NO_FAST=1
run SPMX_NO_COMPRESS=1
