# usage (GPU box): bash scripts/r04_secondary.sh -- the secondary records of the round on the final sources: unsorted C2, the Llama-style
# BPE model, the short forms of C3 / C5, the host forms at three batch sizes, the lattice rates.  gpurun_out/r04s/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-side-configs --no-second-model"
timeout 200 $B --unsorted > $O/bench_uni32k_10m_unsorted.json 2> $O/e1.err
timeout 200 $B --model bpe1k_llama --sentences 4000000 > $O/bench_bpe1k_llama_4m.json 2> $O/e2.err
timeout 200 $B --model bpe32k > $O/bench_bpe32k_10m.json 2> $O/e3.err
timeout 200 $B --model c5_250k --sentences 1000000 --steps 20 --warmup 18 > $O/bench_c5_250k_1m.json 2> $O/e4.err
for N in 10000000 4000000 2000000; do HOST_RATE_ONLY=flat timeout 200 python scripts/host_rate.py $N > $O/host_rate_$N.json 2> $O/h$N.err; done
timeout 200 python scripts/lattice_rate.py 2>/dev/null | tail -1 > $O/lattice_rate.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04s/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    if "ms_per_step" in d: print(f.split("/")[-1], "%.1f M/s %.3f ms" % (d["value"] / 1e6, d["ms_per_step"]), d["roofline"]["all_kernels_ms"])
    elif "flat" in d: print(f.split("/")[-1], "%.1f M sentences/s" % (d["flat"]["sentences_per_s"] / 1e6))
    else: print(f.split("/")[-1], str(d)[:300])
PY
