#!/usr/bin/env python3
"""Rate of the device NBestEncode (spmx_nbest_encode_batch, host-buffer form: H2D + Normalize kernels + NBest kernel +
D2H + host CSR) next to the compiled reference on the host cores.

    python scripts/nbest_rate.py [sentences] [nbest_size] [model]

One JSON line.  The ids of the first 2000 sentences are compared with the oracle's."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from sentencepiece_amd import synth
    from sentencepiece_amd.processor import SentencePieceProcessor
    from tests import oraclelib
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    model = sys.argv[3] if len(sys.argv) > 3 else "uni32k"
    with open(os.path.join(ROOT, "tests", "golden", model + ".model"), "rb") as f:
        blob = f.read()
    sp = SentencePieceProcessor(model_proto=blob, device=0)
    text, offs = synth.ascii_corpus(n, seed=20250227)
    sp.NBestPacked(*synth.gather_packed(text, offs, np.arange(0, n, max(1, n // 1000))), k)      # warm-up
    t0 = time.perf_counter()
    ids, io, sc, ro = sp.NBestPacked(text, offs, k)
    dt = time.perf_counter() - t0
    # parity on a prefix
    o = oraclelib.OracleLib().load(blob)
    fn = o.lib.oracle_nbest_encode
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    tb = text.tobytes()
    ok = True
    t1 = time.perf_counter()
    m = min(n, 2000)
    for i in range(m):
        s = tb[int(offs[i]):int(offs[i + 1])]
        cap = (len(s) + 8) * 4 * k + 64
        out = np.empty(cap, dtype=np.int32)
        oo = np.zeros(k + 2, dtype=np.uint64)
        ss = np.zeros(k + 1, dtype=np.float32)
        r = fn(o.h, s, len(s), k, out.ctypes.data, cap, oo.ctypes.data, ss.ctypes.data)
        got = [ids[int(io[x]):int(io[x + 1])].tolist() for x in range(int(ro[i]), int(ro[i + 1]))]
        want = [out[int(oo[x]):int(oo[x + 1])].tolist() for x in range(r)]
        ok = ok and got == want and np.array_equal(sc[int(ro[i]):int(ro[i + 1])], ss[:r])
    cpu_dt = time.perf_counter() - t1
    print(json.dumps({"metric": "sentences/sec NBestEncode (nbest_size %d), %s, MI355X, host-buffer form" % (k, model),
                      "value": n / dt, "unit": "sentences/s", "ms": dt * 1e3, "sentences": n, "results": int(ro[-1]),
                      "ids": int(io[-1]), "prefix_bit_exact": bool(ok),
                      "oracle_one_core_sentences_per_s": m / cpu_dt}))


if __name__ == "__main__":
    main()
