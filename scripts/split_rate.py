"""Rate of the device corpus packer (spmx_split_lines_device) on a synthetic file image resident in HBM.
Prints one JSON line: GB/s of file bytes, against the algorithmic traffic 2 reads + 1 write + 8 B per line."""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sentencepiece_amd.processor import SentencePieceProcessor   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sentences", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    from tests import fixtures
    sp = SentencePieceProcessor(model_proto=fixtures.model_blob("uni1k"))
    rng = np.random.default_rng(3)
    lens = rng.integers(20, 180, size=a.sentences)
    total = int(lens.sum()) + a.sentences
    img = rng.integers(0x61, 0x7B, size=total, dtype=np.uint8)
    ends = np.cumsum(lens + 1) - 1
    img[ends] = 0x0A
    d_file = torch.zeros(total + 16, dtype=torch.uint8, device="cuda:0")[:total]
    d_file.copy_(torch.from_numpy(img))
    d_text, d_offs, n = sp.SplitLinesDevice(d_file)
    assert n == a.sentences
    want = np.concatenate([[0], np.cumsum(lens)])
    assert np.array_equal(d_offs.cpu().numpy(), want)
    keep = np.ones(total, dtype=bool); keep[ends] = False
    assert np.array_equal(d_text.cpu().numpy(), img[keep])
    lib, h = sp._lib, sp._h
    import ctypes as C
    nn, tb = C.c_uint64(0), C.c_uint64(0)
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    d_text = torch.empty(total, dtype=torch.uint8, device="cuda:0")
    d_offs = torch.empty(n + 1, dtype=torch.int64, device="cuda:0")
    for _ in range(2):
        lib.spmx_split_lines_device(h, d_file.data_ptr(), total, d_text.data_ptr(), total, d_offs.data_ptr(), n + 1, st, C.byref(nn), C.byref(tb))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.steps):
        lib.spmx_split_lines_device(h, d_file.data_ptr(), total, d_text.data_ptr(), total, d_offs.data_ptr(), n + 1, st, C.byref(nn), C.byref(tb))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    alg = 2 * total + (total - n) + 8 * (n + 1)
    print(json.dumps({"metric": "split_lines_file_bytes_per_s", "value": total / ms * 1e3, "unit": "B/s", "file_bytes": total,
                      "lines": n, "ms_per_call": ms, "algorithmic_bytes": alg, "hbm_GBps": alg / ms / 1e6,
                      "frac_of_8TBps": alg / ms / 1e6 / 8000}))


if __name__ == "__main__":
    main()
