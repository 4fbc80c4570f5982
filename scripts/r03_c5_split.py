"""Where the C5 step goes: the 1 M mixed-script sentences split by raw length, each part timed alone (GPU box).
usage: python scripts/r03_c5_split.py [env-free; set SPMX_* outside]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from sentencepiece_amd import synth  # noqa: E402
from sentencepiece_amd.processor import SentencePieceProcessor  # noqa: E402

dev = torch.device("cuda:0")
text, offs = bench.corpus_for("c5_250k", 1_000_000, 20250227, False)
blob = bench.model_blob("c5_250k")
lens = np.diff(offs.astype(np.int64))
print("sentences", len(lens), "bytes", len(text), "max", int(lens.max()), "p50", int(np.median(lens)))
edges = [0, 192, 576, 1536, 4096, 1 << 30]
parts = [("all", np.arange(len(lens)))]
for lo, hi in zip(edges[:-1], edges[1:]):
    parts.append(("(%d, %d]" % (lo, hi), np.nonzero((lens > lo) & (lens <= hi))[0]))
parts.append(("<= 4096", np.nonzero(lens <= 4096)[0]))
parts.append(("<= 1536", np.nonzero(lens <= 1536)[0]))
for name, idx in parts:
    if len(idx) == 0:
        continue
    t, o = synth.gather_packed(text, offs, idx)
    r = bench.side_bench(SentencePieceProcessor, torch, dev, "c5_250k", blob, t, o, 3, 1, name, probe_k=0)
    print("%-14s n=%7d MB=%7.1f maxlen=%6d  %8.3f ms  %s" % (name, len(idx), len(t) / 1e6, int(lens[idx].max()), r["ms_per_step"], r["kernels_ms"]))
    sys.stdout.flush()
