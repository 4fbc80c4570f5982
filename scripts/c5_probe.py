"""C5 (BASELINE configs[4]) under several SPMX_* settings in ONE process on the GPU box: step time, kernel times, the
phase cycles, and every sentence's ids against the compiled reference.
usage: python scripts/c5_probe.py [model] [sentences] -- each further argument "K=V,K=V" is one environment to try"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from sentencepiece_amd.processor import SentencePieceProcessor  # noqa: E402
from tests import fullcheck  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "c5_250k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
envs = sys.argv[3:] or ["SPMX_NO_SPLIT=1", ""]
corpus = os.environ.get("PROBE_CORPUS", "synthetic")
dev = torch.device("cuda:0")
text, offs = bench.corpus_for(model, n, 20250227, False, corpus)
blob = bench.model_blob(model)
d_text = torch.from_numpy(text).to(dev)
d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
check = os.environ.get("PROBE_CHECK", "1") == "1"
for spec in envs:
    kv = dict(x.split("=", 1) for x in spec.split(",") if x)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        sp = SentencePieceProcessor(model_proto=blob, device=0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ids, io, tot = sp.EncodeDevice(d_text, d_offs)
    ids = torch.empty(int(tot) + 64, dtype=torch.int32, device=dev)
    sp.EncodeDevice(d_text, d_offs, ids, io)
    torch.cuda.synchronize()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        sp.EncodeDevice(d_text, d_offs, ids, io)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sp.SetProfiling(True)
    sp.EncodeDevice(d_text, d_offs, ids, io)
    prof = sp.LastProfile()
    sp.SetProfiling(False)
    line = "[%s] %.3f ms/step  %.1f M sentences/s  %.1f GB/s text" % (spec or "default", dt * 1e3, (len(offs) - 1) / dt / 1e6, len(text) / dt / 1e9)
    print(line)
    for c in prof["classes"]:
        if c["kernel"]:
            print("    %-34s %8.3f ms  sentences %9d  cycles %s" % (c["kernel"], c["kernel_ms"], c["sentences"], c["phase_cycles"]))
    print("    path", prof["path"])
    if check:
        io_h = io.cpu().numpy()
        r = fullcheck.compare_all(text, offs, ids[:int(io_h[-1])].cpu().numpy(), io_h, blob, limit_seconds=120.0)
        print("    parity: compared %d, differing %d (first %s), %s, %.1f s" % (r["compared"], r["differing"], r["first"], r["kind"], r["seconds"]))
    sys.stdout.flush()
    del sp
