# usage (GPU box): bash scripts/r04_exp16.sh -- the LDS table of likely words at 4096 entries (a full variant build) against the 2048 of the head.  gpurun_out/r04t/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04t; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs"
timeout 200 $B > $O/bench_hot2k.json 2> $O/a.err
SPMX_LIB=$PWD/sentencepiece_amd/variants/libspmx_hot4k.so timeout 200 $B > $O/bench_hot4k.json 2> $O/b.err
timeout 200 $B > $O/bench_hot2k_b.json 2> $O/c.err
SPMX_LIB=$PWD/sentencepiece_amd/variants/libspmx_hot4k.so timeout 200 $B > $O/bench_hot4k_b.json 2> $O/d.err
python - <<'PY'
import json
for v in ("hot2k", "hot4k", "hot2k_b", "hot4k_b"):
    try:
        d = json.load(open("gpurun_out/r04t/bench_%s.json" % v))
        print(v, "%.3f ms/step" % d["ms_per_step"], d["roofline"]["all_kernels_ms"], "| w16 %.3f ms" % d["long_piece_model"]["ms_per_step"], d["long_piece_model"]["kernels_ms"], d["cpu_baseline"]["probe_ids_bit_exact"] if "cpu_baseline" in d else "")
    except Exception as e:
        print(v, "failed", e)
PY
