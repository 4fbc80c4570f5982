mkdir -p gpurun_out/sw
run() { # name args...
  n=$1; shift
  python bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-second-model --no-side-configs > gpurun_out/sw/$n.json 2> gpurun_out/sw/$n.err
  python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/sw/$n.json').read().strip().splitlines()[-1])
  print('$n', d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline'].get('kernel'))
except Exception as e:
  print('$n failed', e); print(open('gpurun_out/sw/$n.err').read()[-600:])
PY
}
run home
run open --corpus open_vocab
run botchan --corpus botchan
run bpe32k --model bpe32k
run w16 --model uni32k_w16
run llama --model bpe1k_llama
