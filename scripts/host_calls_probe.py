"""spmx_encode_batch (host arrays in and out) call by call, in call order: which calls of bench.py's `end_to_end` are the slow ones.
usage (GPU box): python scripts/host_calls_probe.py [calls]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sentencepiece_amd.processor import SentencePieceProcessor  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 10_000_000
text, offs = bench.corpus_for("uni32k", n, 20250227, False, "synthetic")
sp = SentencePieceProcessor(model_proto=bench.model_blob("uni32k"), device=0)
lib = sp._lib
out = []
for k in range(calls):
    p_ids, p_off = C.c_void_p(), C.c_void_p()
    t0 = time.perf_counter()
    rc = lib.spmx_encode_batch(sp._h, text.ctypes.data, offs.ctypes.data, n, C.byref(p_ids), C.byref(p_off))
    t1 = time.perf_counter()
    assert rc == 0
    lib.spmx_free(p_ids)
    lib.spmx_free(p_off)
    out.append((t1 - t0) * 1e3)
    if os.environ.get("PROBE_SLEEP"):
        time.sleep(float(os.environ["PROBE_SLEEP"]))
print("ms per call, in call order:", " ".join("%.1f" % x for x in out))
