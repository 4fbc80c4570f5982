#!/usr/bin/env python3
"""One-rank RCCL sanity of the multi-GPU output path on a GPU box: init_process_group("nccl", world_size=1), the
IdGatherer with the int16 wire dtype, and encode_sharded end to end against the single-call ids.  (The N > 1 logic is
covered on CPU by tests/test_sharding.py with gloo; this checks the device / RCCL side of the same code.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from sentencepiece_amd import sharding, synth
    from sentencepiece_amd.processor import SentencePieceProcessor
    text, offs = synth.ascii_corpus(200_000, seed=7)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    with open(os.path.join(ROOT, "tests", "golden", "uni32k.model"), "rb") as f:
        sp = SentencePieceProcessor(model_proto=f.read(), device=0)
    d_text = torch.from_numpy(text).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_ids, d_io, total = sp.EncodeDevice(d_text, d_offs)
    g = sharding.IdGatherer(dist, dev, wire_dtype=torch.int16)
    for _ in range(3):                       # steady state: capacities agreed once, async gathers back to back
        g(d_ids, total, d_io)
    parts, poffs = g.result()
    assert torch.equal(parts[0], d_ids[:total]) and torch.equal(poffs[0], d_io)
    full_ids, full_off = sharding.encode_sharded(lambda t, o: sp.EncodeDevice(t, o), text, offs, dist, dev)
    assert torch.equal(full_ids, d_ids[:total]) and torch.equal(full_off, d_io)
    dist.barrier()
    dist.destroy_process_group()
    print("gather sanity ok: %d ids over RCCL (int16 on the wire)" % total)


if __name__ == "__main__":
    main()
