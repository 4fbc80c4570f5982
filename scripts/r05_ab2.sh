# usage (GPU box): bash scripts/r05_ab2.sh <tag>  -- CompactKernel staged through LDS (default) against its search form
# (SPMX_COMPACT_STAGED=0), the coalesced id-offset scan, on C2 and C3; kernel stats of the default; the GPU parity tests
TAG=${1:-r05y}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "%.4g" % d["value"], "%.3f ms" % d["ms_per_step"], "pipeline %.3f" % r.get("pipeline_ms", 0), json.dumps(r.get("all_kernels_ms")))
PY
}
timeout 300 python bench.py --no-cpu-baseline --no-side-configs --no-second-model --steps 5 --warmup 2 > $O/bench_staged.json 2> $O/bench_staged.err; show $O/bench_staged.json
SPMX_COMPACT_STAGED=0 timeout 300 python bench.py --no-cpu-baseline --no-side-configs --no-second-model --steps 5 --warmup 2 > $O/bench_search.json 2> $O/bench_search.err; show $O/bench_search.json
timeout 300 python bench.py --model bpe32k --no-cpu-baseline --no-side-configs --steps 4 --warmup 2 > $O/bench_bpe_staged.json 2> $O/bench_bpe.err; show $O/bench_bpe_staged.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model --no-side-configs > $O/trace_bench.json 2> $O/trace.err
DB=$(find $O/prof -name "x_results.db" | head -1); python scripts/rocpd_summary.py "$DB" > $O/uni32k_10m_kernel_stats.txt 2>&1; rm -rf $O/prof
head -16 $O/uni32k_10m_kernel_stats.txt | cut -c1-150
( time timeout 600 python -m pytest tests/test_word_form.py tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q ) > $O/pytest_gpu_changed.txt 2>&1; tail -4 $O/pytest_gpu_changed.txt
