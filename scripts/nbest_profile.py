"""Where an NBestEncode call's time goes (host arrays in and out): wall time of the call beside the kernels' time that
rocprofv3 --kernel-trace --stats reports for the same process.  Usage (GPU box): python scripts/nbest_profile.py [sentences] [nbest]   (nbest 0: lattice sampling, SampleEncode with nbest_size -1)"""
import json
import sys
import time

sys.path.insert(0, ".")
from sentencepiece_amd import synth                                  # noqa: E402
from sentencepiece_amd.processor import SentencePieceProcessor      # noqa: E402
from tests import fixtures                                           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
text, offs = synth.ascii_corpus(n, seed=7)
sp = SentencePieceProcessor(model_proto=fixtures.model_blob("uni32k"))
call = (lambda: sp.NBestPacked(text, offs, k)) if k > 0 else (lambda: sp.SampleEncodePacked(text, offs, -1, 0.1, seed=1))
call()
ts = []
for _ in range(4):
    t = time.perf_counter()
    call()
    ts.append(time.perf_counter() - t)
print(json.dumps({"sentences": n, "nbest": k, "calls_s": ts, "calls": 5}))
