# Sentence-length differential campaign ON THE GPU (not part of the test suite): the product library's default path -- the
# word-per-lane rounds with the call-local memo, the tail launches, the split form -- and its A/B forms against the compiled
# reference (oracle/_ref) on batches of 30 k - 120 k random sentences over every stored model: words of the model's own
# vocabulary, fresh random words (the call-local memo), words of more than 16 bytes, runs of spaces, punctuation glued to
# words, CJK, accented letters, malformed bytes, empty sentences, a few sentences of several KB.
# usage (GPU box): python scripts/fuzz_sentences_gpu.py SECONDS FIRST_SEED
import gc
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
MODELS = ["uni32k", "bpe32k", "uni32k_w16", "bpe1k_llama", "test_model", "uni1k", "uni1k_bf", "uni1k_ident", "uni1k_suffix",
          "uni1k_uds", "bpe1k", "bpe1k_bf_uds", "bpe1k_noesc", "test_ja_model"]
FORMS = [{}, {"SPMX_WORD_WAVE": "0"}, {"SPMX_NO_DIRECT": "1"}, {"SPMX_NO_WORD_DYN": "1"}, {"SPMX_NO_SPLIT": "1", "SPMX_NO_WORD_NORM": "1"}]
EMU = os.environ.get("FUZZ_EMU") == "1"                 # (a dry run of the script itself on the CPU emulator, tiny batches)
lib = emulib.EmuLib() if EMU else emulib.GpuLib()
ref = refshim.RefLib()
blobs = {m: bench.model_blob(m) for m in MODELS}
refs = {m: ref.load(blobs[m]) for m in MODELS}
# (the product handles of ONE model at a time: a handle keeps its workspaces -- slabs of several GB after a batch with multi-KB
# sentences -- and 70 live handles ran the device out of memory in the script's first version)
wl = synth.WordList()
vocab = [wl.blob[int(o):int(o) + int(l)].tobytes() for o, l in zip(wl.offs[:30000], wl.lens[:30000])]   # uni32k's / bpe32k's own words
corp = fixtures.Corpora()
bot = corp["botchan"][0].tobytes().split()
ja = corp["ja"][0].tobytes().replace(b"\n", b" ")
al = b"abcdefghijklmnopqrstuvwxyzABCDEFXYZ0123456789"
pun = [b",", b".", b"!", b"?", b";", b":", b"'s", b"\"", b")", b"(", b"-", b"--", b"..."]
bad = n_sent = n_bytes = 0
def make_batch(seed):
    rng = np.random.default_rng(seed)
    n = 300 if EMU else int(rng.choice([30000, 60000, 120000]))
    fresh = [bytes(al[int(k)] for k in rng.integers(0, 26, size=int(rng.integers(2, 13)))) for _ in range(int(rng.choice([50, 2000, 40000])))]
    p_fresh = float(rng.choice([0.0, 0.02, 0.1, 0.4]))
    p_odd = float(rng.choice([0.0, 0.01, 0.05]))
    nwords = np.minimum(rng.geometric(1.0 / float(rng.choice([6, 22, 40])), size=n), 600)
    nwords[rng.random(n) < 0.002] = 0                                 # empty sentences
    big = rng.random(n) < 0.0005
    nwords[big] = rng.integers(600, 3000, size=int(big.sum()))     # a few sentences of several KB
    sents = []
    for k in range(n):
        parts = []
        for _ in range(int(nwords[k])):
            r = rng.random()
            if r < p_fresh: w = fresh[int(rng.integers(0, len(fresh)))]
            elif r < p_fresh + p_odd:
                q = rng.random()
                if q < 0.2: w = bytes(al[int(x)] for x in rng.integers(0, len(al), size=int(rng.integers(17, 60))))
                elif q < 0.35: w = "日本語のテキスト処理".encode()[:3 * int(rng.integers(1, 10))]
                elif q < 0.5: w = ("café", "naïve", "über", "ＡＢＣ", "▁x")[int(rng.integers(0, 5))].encode()
                elif q < 0.6: w = bytes([int(rng.integers(0x80, 0x100))])
                elif q < 0.75: w = b" " * int(rng.integers(1, 4))
                elif q < 0.85:
                    a = 3 * int(rng.integers(0, (len(ja) - 90) // 3)); w = ja[a:a + 3 * int(rng.integers(1, 20))]
                else: w = b"\t" if q < 0.9 else b"http://" + bytes(al[int(x)] for x in rng.integers(0, len(al), size=12))
            elif r < 0.55 and vocab: w = vocab[int(rng.integers(0, len(vocab)))]
            else: w = bot[int(rng.integers(0, len(bot)))]
            if rng.random() < 0.08: w = w + pun[int(rng.integers(0, len(pun)))]
            parts.append(w)
        s = b" ".join(parts)
        if rng.random() < 0.01: s = b" " + s
        if rng.random() < 0.01: s = s + b" "
        sents.append(s)
    return synth.pack(sents)


BATCHES = 1 if EMU else 3
while time.time() < t_end:
    batch = []
    for _ in range(BATCHES):
        seed += 1
        batch.append((seed,) + tuple(make_batch(seed)))
        n_sent += len(batch[-1][2]) - 1; n_bytes += len(batch[-1][1])
    for m in MODELS:
        hs = [lib.load(blobs[m], env=e) for e in FORMS]
        for sd, text, offs in batch:
            try:
                ri, ro = refs[m].encode_batch(text, offs)
                for k, h in enumerate(hs):
                    ids, io = h.encode_batch(text, offs)
                    if h.status or not (np.array_equal(io, ro) and np.array_equal(ids, ri)):
                        bad += 1
                        d = np.nonzero(np.diff(io.astype(np.int64)) != np.diff(ro.astype(np.int64)))[0]
                        print("MISMATCH", m, "form", FORMS[k], "seed", sd, h.status, "first sentence with another count", d[:1], flush=True)
            except Exception as e:
                bad += 1; print("EXC", m, sd, repr(e)[:200], flush=True)
        del hs
        gc.collect()
    print("seed", seed, "sentences", n_sent, "MB", n_bytes // 1000000, "x", len(MODELS), "models x", len(FORMS), "forms, bad", bad, flush=True)
print("DONE bad =", bad, "sentences", n_sent, "bytes", n_bytes, "models", len(MODELS), "forms", len(FORMS))
