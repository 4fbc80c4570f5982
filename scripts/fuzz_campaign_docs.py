# Document-length differential campaign on CPU (not part of the test suite): emulator vs oracle on 4-30 KB documents with URL-like and very long words, CJK, malformed bytes.  usage: python scripts/fuzz_campaign_docs.py SECONDS FIRST_SEED
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sentencepiece_amd import synth
from tests import fixtures, oraclelib, emulib
em = emulib.EmuLib(); orc = oraclelib.OracleLib()
t_end = time.time() + float(sys.argv[1])
seed = int(sys.argv[2])
MODELS = ["bpe1k", "bpe1k_llama", "bpe32k", "test_model", "uni1k_bf", "uni1k_ident", "uni1k_uds", "test_ja_model", "uni1k_suffix", "uni32k"]
handles = {m: (em.load(fixtures.model_blob(m)), orc.load(fixtures.model_blob(m))) for m in MODELS}
al = b"abcdefghijklmnopqrstuvwxyz0123456789/_-.%=&?ABCXYZ"
bad = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    words = [b"hello", b"world", b"the", b"tokenizer", b"a", b"of", b"and", b"GPU"]
    docs = []
    for i in range(int(rng.integers(2, 6))):
        parts, ln, target = [], 0, int(rng.integers(4200, 30000))
        while ln < target:
            r = rng.random()
            if r < 0.04: w = b"http://" + bytes(al[int(k)] for k in rng.integers(0, len(al), size=int(rng.integers(18, 120))))
            elif r < 0.045: w = bytes(al[int(k)] for k in rng.integers(0, len(al), size=int(rng.integers(200, 3000))))
            elif r < 0.06: w = "日本語のテキスト処理".encode()[:3 * int(rng.integers(1, 10))]
            elif r < 0.07: w = ("é" * int(rng.integers(1, 60))).encode()
            elif r < 0.075: w = bytes([int(rng.integers(0x80, 0x100))])
            elif r < 0.08: w = "ＡＢＣ　".encode()
            elif r < 0.085: w = b" " * int(rng.integers(1, 5))
            else: w = words[int(rng.integers(0, len(words)))]
            parts.append(w); ln += len(w) + 1
        docs.append(b" ".join(parts))
    text, offs = synth.pack(docs)
    for m in MODELS:
        h, o = handles[m]
        try:
            ids, io = h.encode_batch(text, offs, grid=2)
            oi, oo = o.encode_batch(text, offs)
            if h.status or not (np.array_equal(ids, oi) and np.array_equal(io, oo)):
                bad += 1; print("DOC MISMATCH", m, seed, h.status, flush=True)
        except Exception as e:
            bad += 1; print("EXC", m, seed, repr(e)[:200], flush=True)
    print("seed", seed, "bad", bad, flush=True)
print("DONE bad =", bad)
