# Document-length differential campaign ON THE GPU (not part of the test suite): the product library's wave-cooperative document
# kernels -- one wavefront per document and the walker / folder pair -- against the compiled reference (oracle/_ref) on random
# documents of 4 KB .. 400 KB with URL-like and very long words, CJK, malformed bytes, runs of spaces, over several models.
# usage (GPU box): python scripts/fuzz_docs_gpu.py SECONDS FIRST_SEED
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from sentencepiece_amd import synth
from tests import emulib, fixtures, refshim

t_end = time.time() + float(sys.argv[1])
seed = int(sys.argv[2])
MODELS = ["uni32k", "test_model", "uni1k_bf", "uni1k_ident", "uni1k_suffix", "test_ja_model", "uni32k_w16", "c5_250k_bf"]
lib = emulib.GpuLib()
ref = refshim.RefLib()
handles = {}
for m in MODELS:
    blob = bench.model_blob(m)
    handles[m] = ([lib.load(blob, env={"SPMX_UW_PIPE": p}) for p in ("0", "2", "1")], ref.load(blob))
corp = fixtures.Corpora()
bot = corp["botchan"][0].tobytes().replace(b"\n", b" ")
ja = corp["ja"][0].tobytes().replace(b"\n", b" ")
al = b"abcdefghijklmnopqrstuvwxyz0123456789/_-.%=&?ABCXYZ"
bad = n_docs = n_bytes = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    words = [b"hello", b"world", b"the", b"tokenizer", b"a", b"of", b"and", b"GPU"]
    docs = []
    for i in range(int(rng.integers(3, 40))):
        parts, ln = [], 0
        target = int(rng.choice([4200, 9000, 17000, 40000, 120000, 400000], p=[0.3, 0.25, 0.2, 0.15, 0.07, 0.03]))
        while ln < target:
            r = rng.random()
            if r < 0.04: w = b"http://" + bytes(al[int(k)] for k in rng.integers(0, len(al), size=int(rng.integers(18, 120))))
            elif r < 0.043: w = bytes(al[int(k)] for k in rng.integers(0, len(al), size=int(rng.integers(200, 3000))))
            elif r < 0.06: w = "日本語のテキスト処理".encode()[:3 * int(rng.integers(1, 10))]
            elif r < 0.07: w = ("é" * int(rng.integers(1, 60))).encode()
            elif r < 0.075: w = bytes([int(rng.integers(0x80, 0x100))])
            elif r < 0.08: w = "ＡＢＣ　".encode()
            elif r < 0.085: w = b" " * int(rng.integers(1, 5))
            elif r < 0.25:
                a = int(rng.integers(0, len(bot) - 2000)); w = bot[a:a + int(rng.integers(20, 1500))]
            elif r < 0.30:
                a = 3 * int(rng.integers(0, (len(ja) - 900) // 3)); w = ja[a:a + int(rng.integers(9, 600))]
            else: w = words[int(rng.integers(0, len(words)))]
            parts.append(w); ln += len(w) + 1
        docs.append(b" ".join(parts))
    text, offs = synth.pack(docs)
    n_docs += len(docs); n_bytes += len(text)
    for m in MODELS:
        hs, r = handles[m]
        try:
            ri, ro = r.encode_batch(text, offs)
            for k, h in enumerate(hs):
                ids, io = h.encode_batch(text, offs)
                if h.status or not (np.array_equal(ids, ri) and np.array_equal(io, ro)):
                    bad += 1; print("DOC MISMATCH", m, "form", k, "seed", seed, h.status, flush=True)
        except Exception as e:
            bad += 1; print("EXC", m, seed, repr(e)[:200], flush=True)
    print("seed", seed, "documents", n_docs, "MB", n_bytes // 1000000, "x", len(MODELS), "models x 3 forms, bad", bad, flush=True)
print("DONE bad =", bad, "documents", n_docs, "bytes", n_bytes, "models", len(MODELS))
