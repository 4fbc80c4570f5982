mkdir -p gpurun_out/uw18
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-model --no-side-configs > gpurun_out/uw18/bench_a.json 2> gpurun_out/uw18/bench_a.err
SPMX_EXP_SHARE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-model --no-side-configs > gpurun_out/uw18/bench_b.json 2>> gpurun_out/uw18/bench_a.err
python - <<'PY'
import json
for f in "ab":
    try:
        j=json.loads([l for l in open("gpurun_out/uw18/bench_%s.json"%f) if l.startswith("{")][-1])
        print(f, j["ms_per_step"], j["value"], j["roofline"]["all_kernels_ms"], j["roofline"].get("probe"), j.get("cpu_baseline"))
    except Exception as e: print(f, e)
PY
