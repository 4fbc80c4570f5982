# Near-tie campaign for the unigram word form (kernels_word.h): random unigram models whose scores are QUANTIZED (exact
# ties between segmentations) or differ by a few float ulps (decisions that flip with the magnitude of the accumulated
# score), long all-ASCII sentences built from the models' own words (|best_path_score| up to thousands) -- the device
# kernels under the emulator against the oracle, ids compared one by one.  The word memo's margin guard is what keeps
# the word form bit-exact here; SPMX_WORDMEMO_UNSAFE=1 (a seam of the EMULATOR build, -DSPMX_TEST_SEAMS, that drops the guard) makes the same
# campaign FAIL, which is how one knows it has teeth.
# usage: python scripts/fuzz_wordmemo.py SECONDS FIRST_SEED [--unsafe]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sentencepiece_amd import synth
from tests import fixtures, oraclelib, emulib

from tests.wordfuzz import near_tie_model as make_model, near_tie_corpus as make_corpus   # (shared with tests/test_word_form.py)


def main():
    t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 600)
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    if "--unsafe" in sys.argv:
        os.environ["SPMX_WORDMEMO_UNSAFE"] = "1"
    os.environ["SPMX_WORDMEMO_MIN"] = "1"
    em = emulib.EmuLib(); orc = oraclelib.OracleLib()
    base = fixtures.model_blob("uni1k")
    bad = n_word = n_all = n_models = 0
    while time.time() < t_end:
        seed += 1
        rng = np.random.default_rng(seed)
        blob, words = make_model(rng, base)
        try:
            h = em.load(blob, classes=None)
        except Exception as e:
            print("LOAD", seed, repr(e)[:120], flush=True)
            continue
        o = orc.load(blob)
        n_models += 1
        # the words the memo holds (each encoded alone: the word kernel took the call or it did not); most sentences are
        # built from those only, so that long sentences -- a large accumulated score -- stay in the word form
        hits = []
        for w in sorted(set(words)):
            t1, o1 = synth.pack([w.encode(), w.encode(), b"x" * 40])
            h.encode_batch(t1, o1, grid=1)
            if any(c["kernel"].startswith("EncodeWord") and c["sentences"] >= 2 for c in h.sp.LastProfile()["classes"]):
                hits.append(w)
        if len(hits) < 4:
            continue
        text, offs = make_corpus(rng, hits, 120)
        t2, o2 = make_corpus(rng, words, 40)
        text = np.concatenate([text, t2]); offs = np.concatenate([offs, o2[1:] + offs[-1]])
        ids, io = h.encode_batch(text, offs, grid=2)
        oi, oo = o.encode_batch(text, offs)
        prof = h.sp.LastProfile()
        for c in prof["classes"]:
            if c["kernel"].startswith("EncodeWord"):
                n_word += c["sentences"]
        n_all += len(offs) - 1
        if h.status or not (np.array_equal(io, oo) and np.array_equal(ids, oi)):
            bad += 1
            io_, oo_ = io.astype(np.int64), oo.astype(np.int64)
            k = next((i for i in range(len(offs) - 1) if ids[io_[i]:io_[i + 1]].tolist() != oi[oo_[i]:oo_[i + 1]].tolist()), -1)
            print("MISMATCH seed", seed, "sentence", k, "bytes", int(offs[k + 1] - offs[k]) if k >= 0 else -1, flush=True)
        if n_models % 20 == 0:
            print("seed", seed, "models", n_models, "bad", bad, "word-form share %.3f" % (n_word / max(1, n_all)), flush=True)
    print("DONE models", n_models, "bad", bad, "sentences", n_all, "through the word form", n_word)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
