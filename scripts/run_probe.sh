# usage: bash scripts/run_probe.sh [model] [sentences]  -- one short bench run, compact phase report
M=${1:-uni32k}; N=${2:-2000000}
timeout 300 python bench.py --model $M --sentences $N --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; pc=r['phase_cycles']; print('$M', 'Msent/s %.1f'%(d['value']/1e6), r['all_kernels_ms'], r['kernel'], pc, 'cyc/trip %.0f'%(pc['segment']/max(pc['search_trips'],1)))"
