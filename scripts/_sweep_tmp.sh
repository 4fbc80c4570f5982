mkdir -p gpurun_out/sw
for cfg in hot512:12 hot512:13 hot512:14 hot256:14 hot256:15; do
  v=${cfg%%:*}; w=${cfg##*:}
  export SPMX_LIB=$PWD/sentencepiece_amd/variants/libspmx_$v.so
  SPMX_WORDWAVE_WAVES=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-model --no-side-configs > gpurun_out/sw/$v.$w.json 2> gpurun_out/sw/$v.$w.err
  python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/sw/$v.$w.json').read().strip().splitlines()[-1])
  print('$cfg', d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline'].get('kernel'))
except Exception as e:
  print('$cfg failed', e); print(open('gpurun_out/sw/$v.$w.err').read()[-600:])
PY
done
