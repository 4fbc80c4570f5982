#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer form (spmx_encode_batch): packed text in host memory -> CSR ids in host
memory, timed around the C call only.  Not the headline metric (bench.py times the device-resident form).

    python scripts/host_rate.py [sentences] [model]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sentencepiece_amd import _capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    model = sys.argv[2] if len(sys.argv) > 2 else "uni32k"
    with open(os.path.join(ROOT, "tests", "golden", model + ".model"), "rb") as f:
        blob = f.read()
    lib = _capi.lib()
    h = C.c_void_p()
    assert lib.spmx_create(blob, len(blob), 0, C.byref(h)) == 0
    text, offs = synth.ascii_corpus(n, seed=20250227)
    best = None
    for it in range(4):
        p_ids, p_off = C.c_void_p(), C.c_void_p()
        t0 = time.perf_counter()
        rc = lib.spmx_encode_batch(h, text.ctypes.data, offs.ctypes.data, n, C.byref(p_ids), C.byref(p_off))
        dt = time.perf_counter() - t0
        assert rc == 0, lib.spmx_last_error(h)
        total = int(np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,))[n])
        lib.spmx_free(p_ids)
        lib.spmx_free(p_off)
        if it > 0 and (best is None or dt < best):
            best = dt
    print("host form %s: %d sentences, %.1f MB text in, %.1f MB ids out: %.1f ms -> %.1f M sentences/s, %.2f GB text/s (PCIe-inclusive)"
          % (model, n, len(text) / 1e6, total * 4 / 1e6, best * 1e3, n / best / 1e6, len(text) / best / 1e9))
    lib.spmx_destroy(h)


if __name__ == "__main__":
    main()
