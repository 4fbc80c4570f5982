"""n-best 5 / lattice sampling rate against the HBM budget of the lattice slices (SPMX_NBEST_BUDGET_GB, read at load)."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from sentencepiece_amd import synth
from sentencepiece_amd.processor import SentencePieceProcessor
from tests import fixtures
n = 200000
text, offs = synth.ascii_corpus(n, seed=7)
sp = SentencePieceProcessor(model_proto=fixtures.model_blob("uni32k"))
def rate(fn, reps=3):
    fn(); best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
    return n / best
print(json.dumps({"nbest2": rate(lambda: sp.NBestPacked(text, offs, 2)), "nbest5": rate(lambda: sp.NBestPacked(text, offs, 5)),
                  "nbest16": rate(lambda: sp.NBestPacked(text, offs, 16), 2), "sample_nbest8": rate(lambda: sp.SampleEncodePacked(text, offs, 8, 0.1, seed=1))}))
