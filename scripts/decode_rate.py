#!/usr/bin/env python3
"""Device-resident batch Decode rate (ids -> text), SURVEY.md section 8f row 2.

    python scripts/decode_rate.py [sentences] [model]

Encodes a synthetic corpus on the GPU, then times K decode calls over the resident CSR ids (count pass, scan, write
pass and the two host read-backs included).  Algorithmic bytes per sentence: 4 T' ids + 8 (id offset) read,
L' text bytes + 8 (text offset) written."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from sentencepiece_amd import synth
    from sentencepiece_amd.processor import SentencePieceProcessor
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    model = sys.argv[2] if len(sys.argv) > 2 else "uni32k"
    with open(os.path.join(ROOT, "tests", "golden", model + ".model"), "rb") as f:
        sp = SentencePieceProcessor(model_proto=f.read(), device=0)
    text, offs = synth.ascii_corpus(n, seed=20250227)
    dev = torch.device("cuda", 0)
    d_ids, d_io, total = sp.EncodeDevice(torch.from_numpy(text).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev))
    d_ids = d_ids[:total].clone()
    d_text, d_to, nbytes = sp.DecodeDevice(d_ids, d_io)
    steps = 5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sp.DecodeDevice(d_ids, d_io, d_text, d_to)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    alg = 4 * total + 8 * n + nbytes + 8 * n
    print(json.dumps({"metric": "sentences/sec DecodeBatch, %s, MI355X" % model, "value": n / dt, "unit": "sentences/s",
                      "ms_per_step": dt * 1e3, "gb_text_per_s": nbytes / dt / 1e9,
                      "roofline": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": alg / dt / 8e12, "algorithmic_bytes_per_call": alg},
                      "config": {"workload": "%d sentences, %d ids, %d text bytes, device-resident" % (n, total, nbytes)}}))


if __name__ == "__main__":
    main()
