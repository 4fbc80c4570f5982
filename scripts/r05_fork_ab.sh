# usage (GPU box): bash scripts/r05_fork_ab.sh  -- NOT RUN in round 4 (the GPU budget was spent): how wide the general launch
# beside the word rounds should be when it carries a large share of the work (the Llama-style model,
# 11.5 % of C2's sentences set aside by the scan, the step's critical path at the default 4 wavefronts per workgroup).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_fork; mkdir -p $O
for W in 4 6 8 12; do
  SPMX_FORK_WAVES=$W timeout 200 python bench.py --model bpe1k_llama --sentences 10000000 --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-second-model > $O/llama_fork$W.json 2> $O/llama_fork$W.err
  python - "$O/llama_fork$W.json" $W <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fork_waves", sys.argv[2], "%.1f M sentences/s" % (d["value"] / 1e6), "%.2f ms" % d["ms_per_step"], d["roofline"].get("all_kernels_ms"))
PY
done
