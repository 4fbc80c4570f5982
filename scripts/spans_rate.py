#!/usr/bin/env python3
"""Rates of the forms next to the id-only encode, device-resident:

    python scripts/spans_rate.py [sentences] [model]

  ids      spmx_encode_batch_device                 (the hot path, for scale)
  spans    spmx_encode_batch_spans_device           ids + begin / end (encode with the token-begin column + align kernel)
  normalize spmx_normalize_batch_device             normalized text + norm_to_orig
One JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, steps=5):
    import torch
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    import torch
    from sentencepiece_amd import synth
    from sentencepiece_amd.processor import SentencePieceProcessor
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    model = sys.argv[2] if len(sys.argv) > 2 else "uni32k"
    with open(os.path.join(ROOT, "tests", "golden", model + ".model"), "rb") as f:
        sp = SentencePieceProcessor(model_proto=f.read(), device=0)
    text, offs = synth.ascii_corpus(n, seed=20250227)
    dev = torch.device("cuda", 0)
    d_text = torch.from_numpy(text).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_ids, d_io, total = sp.EncodeDevice(d_text, d_offs)
    t_ids = timed(lambda: sp.EncodeDevice(d_text, d_offs, d_ids, d_io))
    t_spans = timed(lambda: sp.EncodeSpansDevice(d_text, d_offs))
    lib, h = sp._lib, sp._h
    cap = 2 * text.size + 4 * n + 64
    d_norm = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_no = torch.empty(n + 1, dtype=torch.int64, device=dev)
    d_n2o = torch.empty(cap + n + 1, dtype=torch.int32, device=dev)
    tot = C.c_uint64(0)
    st = torch.cuda.current_stream().cuda_stream

    def norm():
        rc = lib.spmx_normalize_batch_device(h, d_text.data_ptr(), d_offs.data_ptr(), n, d_norm.data_ptr(), cap,
                                             d_no.data_ptr(), d_n2o.data_ptr(), st, C.byref(tot))
        assert rc == 0, rc
    t_norm = timed(norm)
    print(json.dumps({"model": model, "sentences": n, "ids": int(total), "raw_bytes": int(text.size),
                      "normalized_bytes": int(tot.value),
                      "ids_sentences_per_s": n / t_ids, "spans_sentences_per_s": n / t_spans,
                      "normalize_sentences_per_s": n / t_norm,
                      "ms": {"ids": t_ids * 1e3, "spans": t_spans * 1e3, "normalize": t_norm * 1e3}}))


if __name__ == "__main__":
    main()
