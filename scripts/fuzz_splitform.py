# Differential campaign for the split form (csrc/kernels_matchfold.h) on CPU, not part of the test suite: the device kernels
# under the emulator against the oracle (and the compiled reference where it is built) -- encode and the spans form -- over
# fresh seeds until the time is up.  Every unigram fixture model the form takes; every length class through it
# (SPMX_SPLIT_MIN=0), the shipped plan (long classes only, beside lane tiles), a launch of split tiles only
# (SPMX_SPLIT_LAUNCH=1) and tight candidate streams (SPMX_SPLIT_CANDS=1: the overflow hand-over).  Inputs: the fuzz corpus of
# tests/test_fuzz.py, mixed-script power-law text up to 4 KB (the classes the form exists for), documents beyond it.
# usage: python scripts/fuzz_splitform.py SECONDS FIRST_SEED
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from sentencepiece_amd import synth  # noqa: E402
from tests import emulib, fixtures, oraclelib  # noqa: E402
from tests.test_fuzz import fuzz_corpus  # noqa: E402

corp = fixtures.Corpora()
em = emulib.EmuLib()
orc = oraclelib.OracleLib()
MODELS = ["test_model", "test_ja_model", "uni1k", "uni1k_bf", "uni1k_ident", "uni1k_suffix", "uni32k", "uni32k_w16", "c5_250k", "c5_250k_bf"]
ENVS = [{"SPMX_SPLIT_MIN": "0"}, {}, {"SPMX_SPLIT_LAUNCH": "1"}, {"SPMX_SPLIT_MIN": "0", "SPMX_SPLIT_CANDS": "1"},
        {"SPMX_SPLIT_MIN": "0", "SPMX_NO_BP_SHORT": "1"}]
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 600)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ref = None
try:
    from tests import refshim
    if refshim.available():
        ref = refshim.RefLib()
except Exception:
    ref = None
bad = n_sent = n_batches = 0
handles = {}
for m in MODELS:
    blob = fixtures.model_blob(m)
    handles[m] = ([em.load(blob, env=dict(e, SPMX_NO_WORD_KERNEL="1"), classes=(None if "SPMX_SPLIT_LAUNCH" in e or not e else emulib.SMALL_CLASSES))
                   for e in ENVS], orc.load(blob), ref.load(blob) if ref else None)
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    ft, fo = fuzz_corpus(60, seed, corp)
    mt, mo = synth.mixed_corpus(int(rng.integers(20, 60)), seed=seed, hi=int(rng.choice([300, 900, 2500, 4096])), sort_by_length=bool(rng.integers(0, 2)))
    lt, lo = synth.mixed_corpus(3, seed=seed + 7, lo=3000, hi=int(rng.choice([4096, 9000, 20000])))
    for m in MODELS:
        hs, o, r = handles[m]
        for text, offs in ((ft, fo), (mt, mo), (lt, lo)):
            oi, oo = o.encode_batch(text, offs)
            if r is not None:
                ri, ro = r.encode_batch(text, offs, threads=2)
                if not (np.array_equal(ri, oi) and np.array_equal(np.asarray(ro).astype(np.int64), np.asarray(oo).astype(np.int64))):
                    bad += 1; print("ORACLE != REFERENCE", m, seed, flush=True)
            want_spans = o.encode_spans(text, offs)
            for k, h in enumerate(hs):
                try:
                    ids, io = h.encode_batch(text, offs)
                    if h.status or not (np.array_equal(ids, oi) and np.array_equal(io, oo)):
                        bad += 1; print("ENCODE MISMATCH", m, seed, ENVS[k], h.status, flush=True)
                    if k < 2:
                        got = h.encode_spans(text, offs)
                        if h.status or not all(np.array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64)) for a, b in zip(got, want_spans)):
                            bad += 1; print("SPANS MISMATCH", m, seed, ENVS[k], h.status, flush=True)
                except Exception as e:
                    bad += 1; print("EXC", m, seed, ENVS[k], repr(e)[:200], flush=True)
                n_sent += len(offs) - 1
                n_batches += 1
    print("seed", seed, "batches", n_batches, "sentences", n_sent, "bad", bad, flush=True)
print("DONE batches %d sentences %d bad = %d" % (n_batches, n_sent, bad))
