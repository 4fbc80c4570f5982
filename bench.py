#!/usr/bin/env python3
"""Headline benchmark: batch encode -> ids on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sentences S] [--model uni32k|bpe32k]

A "step" is one pass of the hot path (classify -> normalize + segment + emit ->
scan -> compact) over one batch of S synthetic sentences (default 10 M, ASCII,
mean 128 B, length-bucketed; 32k unigram model) whose packed text and offsets
are already resident in HBM; ids come out as CSR in HBM.  With N > 1 every rank
encodes its own S sentences (weak scaling, seed + rank) and the step also
all-gathers the id streams of all ranks over RCCL, as BASELINE.json's
north_star asks (``--gather none`` drops the collective).

Prints ONE JSON line on rank 0.  ``roofline`` is for the dominant kernel (the
unigram encode kernel of the busiest length class): algorithmic bytes per
launch (L + 8 + 4 T' + 8 per sentence) over the kernel's
mean duration, measured with HIP events on the launch stream inside the timed
region.  ``cpu_baseline`` times the compiled reference (oracle/_ref, kind
"reference") -- or the plain-C oracle (kind "port") if that is absent -- on a
strided sample of the same corpus on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s


def cpu_baseline(text, offs, model_blob, gpu_counts, gpu_ids=None, gpu_id_offsets=None):
    """Reference CPU path on a bounded strided sample of the bench corpus.  gpu_ids / gpu_id_offsets (host CSR of the
    GPU's output): a 20 k-sentence strided probe is also compared id by id."""
    from sentencepiece_amd import synth
    from tests import refshim
    n = len(offs) - 1
    cores = os.cpu_count() or 1
    if refshim.available():
        h = refshim.RefLib().load(model_blob)
        kind, threads = "reference", cores

        def run(t, o):
            return h.encode_count(t, o, threads=threads)
    else:
        from tests import oraclelib
        h = oraclelib.OracleLib().load(model_blob)
        kind, threads = "port", 1

        def run(t, o):
            return len(h.encode_batch(t, o)[0])
    # calibrate on 20k sentences, then size the sample for ~8 s of wall time
    probe = np.linspace(0, n - 1, num=min(n, 20000)).astype(np.int64)
    pt, po = synth.gather_packed(text, offs, probe)
    t0 = time.perf_counter()
    run(pt, po)
    rate = len(probe) / max(time.perf_counter() - t0, 1e-6)
    s = int(min(n, max(100_000, min(4_000_000, rate * 8.0))))
    pick = np.linspace(0, n - 1, num=s).astype(np.int64)
    st, so = synth.gather_packed(text, offs, pick)
    t0 = time.perf_counter()
    total = run(st, so)
    dt = time.perf_counter() - t0
    out = {"value": s / dt, "unit": "sentences/s", "cores": threads, "kind": kind,
           "sample": "%d sentences strided over the bench corpus (%.1f MB), %.2f s wall, %d host cores present"
                     % (s, len(st) / 1e6, dt, cores),
           "gb_per_s": len(st) / dt / 1e9}
    if gpu_counts is not None:
        out["sample_ids_match_gpu"] = bool(int(gpu_counts[pick].sum()) == int(total))
    if gpu_ids is not None:
        try:
            cids, cio = h.encode_batch(pt, po)
            io = np.asarray(gpu_id_offsets).astype(np.int64)
            lens = (io[1:] - io[:-1])[probe]
            idx = np.repeat(io[:-1][probe] - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(int(lens.sum()))
            out["probe_ids_bit_exact"] = bool(np.array_equal(lens, np.diff(np.asarray(cio).astype(np.int64))) and
                                              np.array_equal(np.asarray(gpu_ids)[idx], np.asarray(cids)))
            out["probe"] = "%d sentences strided over the bench corpus, ids compared one by one" % len(probe)
        except Exception as e:      # the check must not cost the bench line
            out["probe_ids_bit_exact"] = None
            out["probe"] = "failed: %r" % (e,)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sentences", type=int, default=10_000_000, help="sentences per GPU per step")
    ap.add_argument("--model", default="uni32k",
                    help="uni32k | bpe32k (configs[1]/[2], ASCII corpus) | uni32k_w16 (configs[1] with pieces of up to 17 bytes) | "
                         "c5_250k | c5_250k_bf (configs[4], "
                         "250k-piece unigram on the mixed-script power-law corpus)")
    ap.add_argument("--gather", choices=["ids", "none"], default="ids")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unsorted", action="store_true",
                    help="do not length-bucket the synthetic corpus (BASELINE.json's configs are length-bucketed)")
    args = ap.parse_args()

    import torch
    from sentencepiece_amd import synth
    from sentencepiece_amd.processor import SentencePieceProcessor
    from sentencepiece_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # The corpus first: the generator forks a process pool, which is safest before this process has a HIP context
    # or an RCCL communicator.  Weak scaling: every rank draws its own shard of the generator (seed + rank).
    c5 = args.model.startswith("c5_")
    if c5:
        text, offs = synth.mixed_corpus(args.sentences, seed=20250228 + rank)
    else:
        # uni32k_w16: the same generator over words of up to 16 letters (scripts/train_w16.py): pieces of up to 17 bytes,
        # as natural text under the trainer's default max_sentencepiece_length gives -- the generic streaming kernel
        words = synth.WordList(max_word_len=16, mean_word_len=5.5) if args.model.endswith("_w16") else None
        text, offs = synth.ascii_corpus(args.sentences, seed=20250227 + rank, sort_by_length=not args.unsorted, words=words)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if c5:
        from tests import fixtures
        blob = fixtures.model_blob(args.model)      # synthesized 250k-piece model, cached under the temp dir
    else:
        with open(os.path.join(ROOT, "tests", "golden", args.model + ".model"), "rb") as f:
            blob = f.read()
    sp = SentencePieceProcessor(model_proto=blob, device=local)

    n = len(offs) - 1
    d_text = torch.from_numpy(text).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_ids, d_io, total = sp.EncodeDevice(d_text, d_offs)          # sizes the output once
    d_ids = torch.empty(int(total) + 64, dtype=torch.int32, device=dev)
    # ids travel as int16 when the vocabulary allows it (half the bytes on the point-to-point xGMI links)
    wire = torch.int16 if sp.GetPieceSize() <= 32768 else None
    gather = sharding.IdGatherer(dist, dev, wire_dtype=wire) if (world > 1 and args.gather == "ids") else None

    def step():
        _, _, tot = sp.EncodeDevice(d_text, d_offs, d_ids, d_io)
        if gather is not None:
            gather(d_ids, tot, d_io)
        return tot

    for _ in range(args.warmup):
        step()
    if gather is not None:
        gather.wait()
    sp.SetProfiling(True)
    prof = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = step()
        prof.append(sp.LastProfile())
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    sp.SetProfiling(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot_t = torch.tensor([float(len(text)), float(total)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot_t)
        job_bytes, job_ids = float(tot_t[0].item()), float(tot_t[1].item())
    else:
        job_bytes, job_ids = float(len(text)), float(total)

    if rank == 0:
        ms = dt / args.steps * 1e3
        # dominant kernel = the length class with the largest summed kernel time
        ncls = len(prof[0]["classes"])
        k_ms = [sum(p["classes"][c]["kernel_ms"] for p in prof) / len(prof) for c in range(ncls)]
        dom = int(np.argmax(k_ms))
        cls = prof[-1]["classes"][dom]
        achieved = cls["bytes"] / (k_ms[dom] * 1e-3) / 1e9 if k_ms[dom] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        kname = cls["kernel"]
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("%s:%d:%s" % (args.model, args.sentences, kname))
        out = {
            "metric": "sentences/sec EncodeBatch, %s %s, MI355X" % ("250k" if c5 else "32k",
                                                                     "unigram" if sp.model_type() == 1 else "bpe"),
            "value": world * n * args.steps / dt,
            "unit": "sentences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 text -> i32 ids; f64 add / f32 store in the Viterbi relax" if sp.model_type() == 1
                     else "u8 text -> i32 ids; f32 compares",
            "data": "synthetic",
            "gb_text_per_s": job_bytes * args.steps / dt / 1e9,
            "config": {"workload": "configs[%d]: %s model, %d synthetic %s sentences per GPU, mean %.1f B, "
                                   "%s, resident in HBM"
                                   % (4 if c5 else (1 if sp.model_type() == 1 else 2), args.model, n,
                                      "mixed-script power-law [16, 4096] B" if c5 else "ASCII", len(text) / n,
                                      "in generator order (not length-bucketed)" if args.unsorted else "length-bucketed"),
                       "model": args.model, "sentences_per_gpu": n, "ids_per_sentence": job_ids / (world * n),
                       "gather": ("%s (%s on the wire)" % (args.gather, "int16" if wire is not None else "int32")
                                  if args.gather == "ids" else args.gather) if world > 1 else "n/a",
                       "sharding": "dp%d by sentence" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                         "kernel_ms": k_ms[dom], "algorithmic_bytes_per_launch": cls["bytes"],
                         "sentences_per_launch": cls["sentences"],
                         "all_kernels_ms": {prof[-1]["classes"][c]["kernel"]: round(k_ms[c], 4)
                                            for c in range(ncls) if prof[-1]["classes"][c]["kernel"]}, "phase_cycles": cls.get("phase_cycles"), "pipeline_ms": sum(p["total_ms"] for p in prof) / len(prof)},
        }
        if world == 1 and not args.no_cpu_baseline:
            io_h = d_io.cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(text, offs, blob, np.diff(io_h), d_ids[:int(io_h[-1])].cpu().numpy(), io_h)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
