# usage (GPU box): bash scripts/r02_final.sh <tag>  -- the committed head once more: full GPU suite, smoke(), the rates of the neighbouring forms
TAG=${1:-r02z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 300 python scripts/spans_rate.py 2000000 2>/dev/null | tail -1 | tee $O/spans_rate.json
timeout 300 python scripts/decode_rate.py 10000000 2>/dev/null | tail -1 | tee $O/decode_rate.json
timeout 300 python scripts/split_rate.py 2>/dev/null | tail -1 | tee $O/split_rate.json
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-model | tail -c 1500 | tee $O/bench_head.json
