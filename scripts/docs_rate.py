# Document workloads: N documents of D bytes each through the device path with the compiled reference
# timed beside it (one thread, and the best of a thread sweep) on the same box; ids of a few documents compared one by one.
# usage (GPU box): python scripts/docs_rate.py [--model uni32k] [--docs 8192] [--bytes 16384] [--steps 3]
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def make_docs(n_docs, doc_bytes, seed=7):
    """Documents of ~doc_bytes: sentences of the C2 generator (in generator order) joined by single spaces."""
    from sentencepiece_amd import synth
    per = max(1, doc_bytes // 127)
    text, offs = synth.ascii_corpus(n_docs * per, seed=seed, sort_by_length=False)
    offs = offs.astype(np.int64)
    inner = np.ones(len(offs) - 1, dtype=bool)
    inner[per - 1::per] = False                       # no separator after a document's last sentence
    ins = offs[1:][inner]
    out = np.insert(text, ins, 0x20)
    shift = np.concatenate([[0], np.cumsum(inner)])
    doc_offs = (offs + shift)[::per].astype(np.uint64)
    if len(doc_offs) != n_docs + 1:
        doc_offs = np.concatenate([doc_offs, [len(out)]]).astype(np.uint64)[:n_docs + 1]
    return np.ascontiguousarray(out), doc_offs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="uni32k")
    ap.add_argument("--docs", type=int, default=8192)
    ap.add_argument("--bytes", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    a = ap.parse_args()
    import torch
    from sentencepiece_amd.processor import SentencePieceProcessor
    from tests import fixtures, refshim
    text, offs = make_docs(a.docs, a.bytes)
    blob = fixtures.model_blob(a.model)
    sp = SentencePieceProcessor(model_proto=blob)
    dev = torch.device("cuda", 0)
    d_text = torch.from_numpy(text).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_ids, d_io, total = sp.EncodeDevice(d_text, d_offs)
    d_ids = torch.empty(int(total) + 64, dtype=torch.int32, device=dev)
    sp.EncodeDevice(d_text, d_offs, d_ids, d_io)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sp.EncodeDevice(d_text, d_offs, d_ids, d_io)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    sp.SetProfiling(True)
    sp.EncodeDevice(d_text, d_offs, d_ids, d_io)
    prof = sp.LastProfile()
    out = {"workload": "%d documents of ~%d bytes (C2 sentences joined by spaces), %s, resident in HBM" % (a.docs, a.bytes, a.model),
           "model": a.model, "docs": a.docs, "doc_bytes": a.bytes, "total_mb": len(text) / 1e6, "gpu_seconds": dt,
           "gpu_mb_per_s": len(text) / 1e6 / dt, "gpu_docs_per_s": a.docs / dt, "ids": int(total),
           "kernels_ms": {c["kernel"]: round(c["kernel_ms"], 3) for c in prof["classes"] if c["kernel"]}}
    if refshim.available():
        from sentencepiece_amd import synth
        h = refshim.RefLib().load(blob)
        io = d_io.cpu().numpy().astype(np.int64)
        ids = d_ids[:int(io[-1])].cpu().numpy()
        # parity: a few documents, id by id
        pick = np.unique(np.linspace(0, a.docs - 1, num=min(a.docs, 6)).astype(np.int64))
        pt, po = synth.gather_packed(text, offs, pick)
        cids, cio = h.encode_batch(pt, po)
        got = np.concatenate([ids[io[i]:io[i + 1]] for i in pick])
        out["probe_ids_bit_exact"] = bool(np.array_equal(got, np.asarray(cids)))
        # the reference on this box's host cores: one thread, then the best of a sweep, on a bounded sample of the documents
        cores = os.cpu_count() or 1

        def rate(threads, seconds):
            k = max(threads, 1)
            k = min(a.docs, k)
            pick = np.linspace(0, a.docs - 1, num=k).astype(np.int64)
            pt, po = synth.gather_packed(text, offs, pick)
            t0 = time.perf_counter()
            h.encode_count(pt, po, threads=threads)
            d = time.perf_counter() - t0
            reps = max(1, int(seconds / max(d, 1e-3)))
            if reps > 1 and k < a.docs:
                k2 = min(a.docs, k * reps)
                pick = np.linspace(0, a.docs - 1, num=k2).astype(np.int64)
                pt, po = synth.gather_packed(text, offs, pick)
                t0 = time.perf_counter()
                h.encode_count(pt, po, threads=threads)
                d = time.perf_counter() - t0
            return len(pt) / 1e6 / d
        out["cpu_one_thread_mb_per_s"] = rate(1, a.cpu_seconds)
        sweep = {str(t): rate(t, a.cpu_seconds / 2) for t in (16, 32, 64, 128) if t <= max(cores, 16)}
        out["cpu_thread_sweep_mb_per_s"] = sweep
        out["cpu_best_mb_per_s"] = max(sweep.values())
        out["gpu_vs_cpu_best"] = out["gpu_mb_per_s"] / out["cpu_best_mb_per_s"]
        out["host_cores"] = cores
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
