#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer forms -- never the headline metric (bench.py times the device-resident
form) -- on the C2 corpus:

  flat      spmx_encode_batch(_ex): packed text + offsets in host memory -> CSR ids in host memory (chunk pipeline)
  views     spmx_encode_batch_views: n (pointer, length) pairs, what EncodeBatch(vector<string_view>) passes
  nested    include/spmx_processor.h EncodeBatch(vector<string_view>) -> vector<vector<int>>, timed in C++ (tools/host_bench.cc)
  python    sp.encode(list[str]) -> list[list[int]] (1 M sentences)
  file      spmx_encode_file: corpus file -> flat binary ids

    python scripts/host_rate.py [sentences] [model]        (one JSON line)
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sentencepiece_amd import _capi, synth  # noqa: E402
from sentencepiece_amd.processor import SentencePieceProcessor  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    model = sys.argv[2] if len(sys.argv) > 2 else "uni32k"
    mpath = os.path.join(ROOT, "tests", "golden", model + ".model")
    sp = SentencePieceProcessor(model_file=mpath)
    lib = _capi.lib()
    text, offs = synth.ascii_corpus(n, seed=20250227)
    out = {"sentences": n, "model": model, "text_mb": len(text) / 1e6, "host_threads": int(os.environ.get("SPMX_HOST_THREADS", "0")), "host_chunk": int(os.environ.get("SPMX_HOST_CHUNK", "0"))}

    def timed(fn, reps=4):
        best = None
        for it in range(reps):
            t0 = time.perf_counter()
            r = fn()
            dt = time.perf_counter() - t0
            if it > 0 and (best is None or dt < best):
                best = dt
        return best, r

    def flat():
        p_ids, p_off = C.c_void_p(), C.c_void_p()
        rc = lib.spmx_encode_batch(sp._h, text.ctypes.data, offs.ctypes.data, n, C.byref(p_ids), C.byref(p_off))
        assert rc == 0, lib.spmx_last_error(None)
        total = int(np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,))[n])
        lib.spmx_free(p_ids)
        lib.spmx_free(p_off)
        return total
    dt, total = timed(flat)
    out["flat"] = {"ms": dt * 1e3, "sentences_per_s": n / dt, "gb_text_per_s": len(text) / dt / 1e9, "ids": total}
    if os.environ.get("HOST_RATE_ONLY") == "flat":
        print(json.dumps(out))
        return

    # the Python list form on a slice
    m = min(n, 1_000_000)
    tb = text[:int(offs[m])].tobytes()
    strs = [tb[int(offs[i]):int(offs[i + 1])].decode("utf-8", "replace") for i in range(m)]
    dt, _ = timed(lambda: sp.encode(strs), reps=3)
    out["python_list"] = {"sentences": m, "ms": dt * 1e3, "sentences_per_s": m / dt}
    dt, _ = timed(lambda: sp.EncodeAsArrays(strs), reps=3)
    out["python_arrays"] = {"sentences": m, "ms": dt * 1e3, "sentences_per_s": m / dt,
                            "what": "sp.EncodeAsArrays(list[str]) -> (ids, id_offsets) numpy arrays"}
    t0 = time.perf_counter()
    for k in range(200):
        sp.encode(strs[(k * 7919) % m])
    out["python_single_encode_us"] = (time.perf_counter() - t0) / 200 * 1e6

    # the C++ facade and the file tool, on the corpus as a file
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "corpus.txt")
        lens = np.diff(offs.astype(np.int64))
        buf = np.full(len(text) + n, 0x0A, dtype=np.uint8)
        pos = offs[:-1].astype(np.int64) + np.arange(n)
        idx = np.repeat(pos - offs[:-1].astype(np.int64), lens) + np.arange(len(text))
        buf[idx] = text
        buf.tofile(path)
        exe = os.path.join(td, "host_bench")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tools", "host_bench.cc"),
                               "-L" + os.path.join(ROOT, "sentencepiece_amd"), "-lspmx", "-Wl,-rpath," + os.path.join(ROOT, "sentencepiece_amd")])
        r = subprocess.run([exe, mpath, path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out["cpp"] = json.loads(r.stdout.strip().splitlines()[-1])
        best = None
        for it in range(3):
            t0 = time.perf_counter()
            ns, ni = sp.EncodeFile(path, os.path.join(td, "ids.bin"), "bin")
            dt = time.perf_counter() - t0
            if it > 0 and (best is None or dt < best):
                best = dt
        out["file_bin"] = {"ms": best * 1e3, "sentences_per_s": ns / best, "gb_text_per_s": os.path.getsize(path) / best / 1e9, "ids": ni}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
