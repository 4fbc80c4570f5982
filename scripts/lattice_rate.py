"""Rates of the lattice entry points (NBestEncode, SampleEncode, the kOriginal encoder) and of BPE-dropout on the GPU:
host CSR in, host CSR out (these forms have no device-resident entry).  Usage: python scripts/lattice_rate.py [sentences]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sentencepiece_amd import synth                                  # noqa: E402
from sentencepiece_amd.processor import SentencePieceProcessor      # noqa: E402
from tests import fixtures                                           # noqa: E402


def rate(fn, n, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t)
    return n / best


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    text, offs = synth.ascii_corpus(n, seed=7)
    out = {"sentences": n, "mean_bytes": float(offs[-1]) / n}
    sp = SentencePieceProcessor(model_proto=fixtures.model_blob("uni32k"))
    out["encode_host"] = rate(lambda: sp.EncodePacked(text, offs), n)
    for k in (2, 5, 16, 64):
        m = n if k <= 5 else n // 8
        t2, o2 = text[:int(offs[m])], offs[:m + 1]
        out["nbest%d" % k] = rate(lambda: sp.NBestPacked(t2, o2, k), m)
    m8 = n // 8
    t8, o8 = text[:int(offs[m8])], offs[:m8 + 1]
    out["nbest5_eighth_of_the_batch"] = rate(lambda: sp.NBestPacked(t8, o8, 5), m8)     # (16 and 64 above run on this eighth)
    out["nbest5_spans"] = rate(lambda: sp.NBestSpansPacked(text, offs, 5), n)
    out["sample_lattice"] = rate(lambda: sp.SampleEncodePacked(text, offs, -1, 0.1, seed=1), n)
    out["sample_nbest8"] = rate(lambda: sp.SampleEncodePacked(text, offs, 8, 0.1, seed=1), n)
    out["original_viterbi"] = rate(lambda: sp.EncodeOriginalPacked(text, offs), n)
    bp = SentencePieceProcessor(model_proto=fixtures.model_blob("bpe32k"))
    out["bpe_encode_host"] = rate(lambda: bp.EncodePacked(text, offs), n)
    out["bpe_dropout_0.1"] = rate(lambda: bp.SampleEncodePacked(text, offs, -1, 0.1, seed=1), n)
    out["unit"] = "sentences/s, host arrays in and out"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
