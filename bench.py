#!/usr/bin/env python3
"""Headline benchmark: batch encode -> ids on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sentences S] [--model uni32k|bpe32k|...]

A "step" is one pass of the hot path (classify -> normalize + segment + emit -> scan -> compact) over one batch of S
synthetic sentences (default 10 M, ASCII, mean 128 B, length-bucketed; 32k unigram model) whose packed text and offsets
are already resident in HBM; ids come out as CSR in HBM.  With N > 1 every rank encodes its own S sentences (weak
scaling, seed + rank; default 12.5 M per rank = configs[3] at N = 8) and the step also all-gathers the id streams of
all ranks over RCCL, as BASELINE.json's north_star asks; the same run then times the no-collective form too
(``value_gather_none``).

Prints ONE JSON line on rank 0.
  value         sentences/s of K unprofiled steps (barrier + synchronize on both sides, MAX over ranks)
  roofline      the dominant kernel (the streaming encode launch) over a second, PROFILED loop of the same steps: HIP
                events around the launch on its stream; algorithmic bytes = L + 8 + 4 T' + 8 per sentence;
                `traffic` = HBM bytes per launch from the rocprofv3 PMC passes kept in profiles/pmc_traffic.json -- used
                only if that file was made from the kernel sources this run was built from (a hash of csrc/), else null
  long_piece_model   (N = 1, unigram headline) the same corpus recipe with words of up to 16 letters and a model trained
                on it (pieces of up to 17 bytes, what the trainer's default max_sentencepiece_length gives on natural
                text): the generic streaming kernel instead of the 16-entry-ring specialization
  cpu_baseline  the compiled reference (oracle/_ref, kind "reference") on this box's host cores, bounded samples of the
                same corpus: one thread, a thread sweep (best reported), and `spm_encode` file -> file on configs[0]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s


def kernel_sources_sha():
    """Hash of the device sources: ties profiles/pmc_traffic.json to the kernels it was measured on."""
    d = os.path.join(ROOT, "sentencepiece_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def traffic_on_record(model, sentences, kernel):
    """(bytes, bytes_uncorrected, note) of the PMC pass on record for this model / size / kernel (profiles/pmc_traffic.json,
    made by scripts/pmc_traffic.sh) -- used only when it was measured on exactly these kernel sources."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    sha = kernel_sources_sha()
    if not os.path.exists(tpath):
        return None, None, "no PMC pass on record for this model / size / kernel"
    with open(tpath) as f:
        recs = json.load(f)
    rec = recs.get("%s:%d:%s" % (model, sentences, kernel))
    if not isinstance(rec, dict):      # (a kernel compiled for several row lengths: the profile names the instance, "UniLongKernel<16u>")
        inst = [v for k, v in recs.items() if k.startswith("%s:%d:%s<" % (model, sentences, kernel)) and isinstance(v, dict)]
        rec = inst[0] if len(inst) == 1 else None
    if not isinstance(rec, dict):
        return None, None, "no PMC pass on record for this model / size / kernel"
    if rec.get("src_sha") != sha:
        return None, None, "profiles/pmc_traffic.json is from other kernel sources (%s, now %s): not used" % (rec.get("src_sha"), sha)
    return rec.get("bytes"), rec.get("bytes_lower"), rec.get("note", "profiles/pmc_traffic.json, made from these kernel sources")


def cpu_baseline(text, offs, model_blob, gpu_counts, gpu_ids=None, gpu_id_offsets=None):
    """Reference CPU path on bounded strided samples of the bench corpus (about 20 s in all)."""
    from sentencepiece_amd import synth
    from tests import refshim
    n = len(offs) - 1
    cores = os.cpu_count() or 1
    if not refshim.available():
        from tests import oraclelib
        h = oraclelib.OracleLib().load(model_blob)
        pick = np.linspace(0, n - 1, num=min(n, 200_000)).astype(np.int64)
        st, so = synth.gather_packed(text, offs, pick)
        t0 = time.perf_counter()
        h.encode_batch(st, so)
        dt = time.perf_counter() - t0
        return {"value": len(pick) / dt, "unit": "sentences/s", "cores": 1, "kind": "port",
                "sample": "%d sentences strided over the bench corpus, plain-C restatement on one thread" % len(pick)}
    h = refshim.RefLib().load(model_blob)

    def rate(threads, seconds):
        """sentences/s of the reference's Encode loop on `threads` threads over a sample sized for ~`seconds`."""
        probe = np.linspace(0, n - 1, num=min(n, 4000 * max(1, threads // 8))).astype(np.int64)
        pt, po = synth.gather_packed(text, offs, probe)
        t0 = time.perf_counter()
        h.encode_count(pt, po, threads=threads)
        r0 = len(probe) / max(time.perf_counter() - t0, 1e-6)
        s = int(min(n, max(20_000, r0 * seconds)))
        pick = np.linspace(0, n - 1, num=s).astype(np.int64)
        st, so = synth.gather_packed(text, offs, pick)
        t0 = time.perf_counter()
        total = h.encode_count(st, so, threads=threads)
        dt = time.perf_counter() - t0
        return s / dt, s, dt, pick, int(total), len(st)
    one, s1, dt1, _, _, _ = rate(1, 3.0)
    sweep = {}
    best = (0.0, 1, 0, 0.0, None, 0, 0)
    for t in [t for t in (16, 32, 64, 128, 256) if t <= max(cores, 16)]:
        r, s, dt, pick, total, nbytes = rate(t, 2.0)
        sweep[str(t)] = r
        if r > best[0]:
            best = (r, t, s, dt, pick, total, nbytes)
    out = {"value": best[0], "unit": "sentences/s", "cores": best[1], "kind": "reference",
           "sample": "%d sentences strided over the bench corpus (%.1f MB), %.2f s wall on %d threads (best of the sweep); "
                     "%d host cores present" % (best[2], best[6] / 1e6, best[3], best[1], cores),
           "one_thread": {"value": one, "sentences": s1, "seconds": dt1,
                          "what": "in-process loop over pre-loaded strings calling Encode(s, &ids) on one thread"},
           "thread_sweep": sweep,
           "scheme": "atomic-counter workers over pre-loaded strings, as python/src/sentencepiece/sentencepiece.i:245-267"}
    if gpu_counts is not None and best[4] is not None:
        out["sample_ids_match_gpu"] = bool(int(gpu_counts[best[4]].sum()) == best[5])
    # configs[0]: spm_encode --output_format=id, file -> file, one thread (the reference's own command line)
    exe = os.path.join(ROOT, "oracle", "_ref", "spm_encode")
    bot = os.path.join(ROOT, "tests", "golden", "botchan.txt")
    mdl = os.path.join(ROOT, "tests", "golden", "test_model.model")
    if os.path.exists(exe):
        try:
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                best_c1 = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    subprocess.check_call([exe, "--model=" + mdl, "--output_format=id", "--output=" + os.path.join(td, "o.txt"), bot],
                                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    dt = time.perf_counter() - t0
                    best_c1 = dt if best_c1 is None else min(best_c1, dt)
                with open(os.path.join(td, "o.txt"), "rb") as f:
                    md5 = hashlib.md5(f.read()).hexdigest()
            out["spm_encode_c1"] = {"what": "oracle/_ref/spm_encode --output_format=id on botchan.txt (4288 lines, 1k unigram), file -> file, "
                                            "process start and model load included", "seconds": best_c1, "sentences_per_s": 4288 / best_c1,
                                    "md5": md5}
        except Exception as e:      # the check must not cost the bench line
            out["spm_encode_c1"] = {"failed": repr(e)}
    if gpu_ids is not None:
        try:
            out["probe_ids_bit_exact"], out["probe"] = probe_exact(text, offs, model_blob, gpu_ids, gpu_id_offsets)
        except Exception as e:
            out["probe_ids_bit_exact"] = None
            out["probe"] = "failed: %r" % (e,)
    return out


def probe_exact(text, offs, model_blob, gpu_ids, gpu_id_offsets, k=None):
    """EVERY sentence's ids against the compiled reference (tests/fullcheck.py: its Encode loop on all host cores over
    chunks of the corpus; the oracle where the reference is not built).  k: only a strided sample of k sentences.
    -> (bit_exact or None, text of what was compared)"""
    from sentencepiece_amd import synth
    from tests import fullcheck
    n = len(offs) - 1
    if k is not None and k < n:
        probe = np.unique(np.linspace(0, n - 1, num=k).astype(np.int64))
        pt, po = synth.gather_packed(text, offs, probe)
        io = np.asarray(gpu_id_offsets).astype(np.int64)
        lens = (io[1:] - io[:-1])[probe]
        idx = np.repeat(io[:-1][probe] - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(int(lens.sum()))
        sub_io = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        r = fullcheck.compare_all(pt, po, np.asarray(gpu_ids)[idx], sub_io, model_blob)
        what = "%d sentences strided over the %d of the batch" % (len(probe), n)
    else:
        r = fullcheck.compare_all(text, offs, gpu_ids, gpu_id_offsets, model_blob, limit_seconds=90.0)
        what = "%d sentences = the whole batch" % r["compared"] if r["compared"] == n else \
               "%d of the batch's %d sentences (time limit)" % (r["compared"], n)
    return r["differing"] == 0, "%s, ids compared one by one with the %s in %.1f s: %d sentences differ" % (
        what, r["kind"], r["seconds"], r["differing"])


def side_bench(sp_cls, torch, dev, name, blob, text, offs, steps, warmup, what, probe_k=None, corpus=None):
    """A compact record of one more single-GPU configuration: value, kernels, roofline fraction of
    the dominant kernel, ids of a sample against the compiled reference."""
    sp = sp_cls(model_proto=blob, device=dev.index or 0)
    d_text = torch.from_numpy(text).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    ids, io, tot = sp.EncodeDevice(d_text, d_offs)
    ids = torch.empty(int(tot) + 64, dtype=torch.int32, device=dev)
    for _ in range(warmup):
        sp.EncodeDevice(d_text, d_offs, ids, io)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sp.EncodeDevice(d_text, d_offs, ids, io)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sp.SetProfiling(True)
    sp.EncodeDevice(d_text, d_offs, ids, io)
    prof = sp.LastProfile()
    sp.SetProfiling(False)
    n = len(offs) - 1
    cls = [c for c in prof["classes"] if c["kernel"]]
    dom = max(cls, key=lambda c: c["bytes"]) if cls else None        # (the launch that does most of the work, as in the headline)
    out = {"model": name, "what": what, "value": n / dt, "unit": "sentences/s", "ms_per_step": dt * 1e3,
           "gb_text_per_s": len(text) / dt / 1e9, "sentences": n, "mean_bytes": len(text) / max(n, 1),
           "ids_per_sentence": float(tot) / max(n, 1),
           "kernels_ms": {c["kernel"]: round(c["kernel_ms"], 4) for c in cls}}
    if dom is not None and dom["kernel_ms"] > 0:
        ach = dom["bytes"] / (dom["kernel_ms"] * 1e-3) / 1e9
        out["roofline"] = {"kernel": dom["kernel"], "kernel_ms": dom["kernel_ms"], "sentences_per_launch": dom["sentences"],
                           "algorithmic_bytes_per_launch": dom["bytes"], "achieved": ach, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS}
        rec_name = name if corpus is None else name + "@" + corpus
        out["roofline"]["traffic"], _, out["roofline"]["traffic_note"] = traffic_on_record(rec_name, n, dom["kernel"])
        # The launch that takes the LONGEST may complete few sentences (a collecting round whose sentences finish in the
        # next one, a round that waits for CUs another launch holds): it is named beside the dominant one with what it
        # completed, so that roofline.frac cannot be read as the step's.
        lng = max(cls, key=lambda c: c["kernel_ms"])
        if lng is not dom and lng["kernel_ms"] > 0:
            l_ach = lng["bytes"] / (lng["kernel_ms"] * 1e-3) / 1e9
            out["roofline"]["longest_kernel"] = {"kernel": lng["kernel"], "kernel_ms": lng["kernel_ms"],
                                                 "sentences_per_launch": lng["sentences"],
                                                 "algorithmic_bytes_per_launch": lng["bytes"], "achieved": l_ach,
                                                 "frac": l_ach / HBM_PEAK_GBS,
                                                 "traffic": traffic_on_record(rec_name, n, lng["kernel"])[0]}
        # the step as a whole: every launch's algorithmic bytes over the step's wall time (scan, compact, gaps included)
        p_bytes = sum(c["bytes"] for c in cls)
        p_ach = p_bytes / dt / 1e9
        k_traffic = [traffic_on_record(rec_name, n, c["kernel"])[0] for c in cls]
        out["pipeline"] = {"algorithmic_bytes": p_bytes, "ms": dt * 1e3, "achieved": p_ach, "unit": "GB/s",
                           "frac": p_ach / HBM_PEAK_GBS,
                           "traffic_of_the_encode_launches": sum(t for t in k_traffic if t is not None) if any(t is not None for t in k_traffic) else None,
                           "launches_without_a_traffic_record": [c["kernel"] for c, t in zip(cls, k_traffic) if t is None],
                           "what": "every launch's algorithmic bytes over the whole step (classify, scan, compact included in the time)"}
    try:
        io_h = io.cpu().numpy()
        out["probe_ids_bit_exact"], out["probe"] = probe_exact(text, offs, blob, ids[:int(io_h[-1])].cpu().numpy(), io_h, probe_k)
    except Exception as e:      # the check must not cost the bench line
        out["probe_ids_bit_exact"] = None
        out["probe_error"] = repr(e)[:200]
    del sp, d_text, d_offs, ids, io
    return out


GATHER_DEPTH = 3          # gathers in flight (sharding.IdGatherer): batch k's transfer runs under the encodes of k + 1 and k + 2
XGMI_LINKS = 7            # peers one hop away on an 8-GPU MI355X node (point to point, no switch)
XGMI_LINK_GBS = 76.5      # one direction of one link: 153 GB/s bidirectional


def gather_bound(world, sentences_per_rank, ids_per_sentence, wire_bytes, count_bytes, encode_ms):
    """What the north star's all-gather of the ids costs on the node's links, from numbers the run has: every rank sends its
    ids (wire_bytes each) and per-sentence counts (count_bytes each) to every peer, one link per peer, so a link carries one
    rank's payload per step in each direction; with the transfer of batch k under the encode of batch k + 1 the step is
    the longer of the two.  A MODEL (link peak, perfect overlap) -- the measured figures are value_gather_* ."""
    payload = sentences_per_rank * (ids_per_sentence * wire_bytes + count_bytes)
    peers = max(world - 1, 1)
    link_ms = payload / (XGMI_LINK_GBS * 1e9) * 1e3
    step_ms = max(encode_ms, link_ms)
    return {"world": world, "payload_bytes_per_rank": payload, "ids_wire_bytes": wire_bytes, "count_wire_bytes": count_bytes,
            "ingest_bytes_per_rank_per_step": payload * peers, "encode_ms": encode_ms,
            "ingest_gb_per_s_needed_to_hide_it": payload * peers / (encode_ms * 1e-3) / 1e9,
            "links": XGMI_LINKS, "link_gb_per_s_one_direction": XGMI_LINK_GBS,
            "ingest_gb_per_s_available": XGMI_LINK_GBS * min(peers, XGMI_LINKS),
            "gather_ms_at_link_peak": link_ms, "predicted_step_ms": step_ms,
            "predicted_scaling_vs_one_gpu": world * encode_ms / step_ms,
            "bound": "links (the gather)" if link_ms > encode_ms else "the encode",
            "payload_bytes_per_rank_for_6x_of_8": XGMI_LINK_GBS * 1e9 * encode_ms * 1e-3 * 8.0 / 6.0,
            "what": "model: every rank's payload crosses one xGMI link per peer per step at the link's one-direction peak, "
                    "fully overlapped with the next batch's encode; world = 8 is a projection from this run's own rate when "
                    "the run has fewer ranks"}


def corpus_for(model, sentences, seed, unsorted, corpus="synthetic"):
    from sentencepiece_amd import synth
    if corpus == "open_vocab":
        return synth.open_vocab_corpus(sentences, seed=20250301)
    if corpus == "botchan":
        return synth.repeated_file_corpus(os.path.join(ROOT, "tests", "golden", "botchan.txt"), 2000)
    if corpus in ("docs_16k", "docs_1m"):
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import docs_rate
        return docs_rate.make_docs(*((8192, 16384) if corpus == "docs_16k" else (256, 1 << 20)))
    if model.startswith("c5_"):
        return synth.mixed_corpus(sentences, seed=seed + 1)
    # uni32k_w16: the same generator over words of up to 16 letters (scripts/train_w16.py): pieces of up to 17 bytes,
    # as natural text under the trainer's default max_sentencepiece_length gives -- the generic streaming kernel
    words = synth.WordList(max_word_len=16, mean_word_len=5.5) if model.endswith("_w16") else None
    return synth.ascii_corpus(sentences, seed=seed, sort_by_length=not unsorted, words=words)


def model_blob(model):
    if model.startswith("c5_"):
        from tests import fixtures
        return fixtures.model_blob(model)      # synthesized 250k-piece model, cached under the temp dir
    with open(os.path.join(ROOT, "tests", "golden", model + ".model"), "rb") as f:
        return f.read()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU) of ONE node.  Launched under torch.distributed.run it must equal WORLD_SIZE; "
                         "launched plainly with N > 1 the script starts its N ranks itself (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sentences", type=int, default=None,
                    help="sentences per GPU per step (default 10 M; 12.5 M with --gpus > 1: configs[3] is 100 M over 8 GPUs)")
    ap.add_argument("--model", default="uni32k",
                    help="uni32k | bpe32k (configs[1]/[2], ASCII corpus) | uni32k_w16 (configs[1] with pieces of up to 17 bytes) | "
                         "c5_250k | c5_250k_bf (configs[4], 250k-piece unigram on the mixed-script power-law corpus)")
    ap.add_argument("--gather", choices=["both", "ids", "none"], default="both",
                    help="N > 1: all-gather the ids over RCCL (the north star), leave it out, or time both (default)")
    ap.add_argument("--gather-algo", choices=["both", "all_gather", "p2p", "p2p_exact"], default="both",
                    help="how the ids travel: the library's all-gather (every rank padded to the largest capacity), or exact-size "
                         "point-to-point sends (world - 1 per rank, posted as one batch); default: time BOTH in this run "
                         "(sharding.IdGatherer algo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-model", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="leave the c3 (32k BPE), c5 (250k unigram, mixed script) and document sub-records out of the line")
    ap.add_argument("--corpus", choices=["synthetic", "open_vocab", "botchan", "docs_16k", "docs_1m"], default="synthetic",
                    help="one of the side configurations' corpora as the main workload (scripts/pmc_traffic.sh: the PMC passes "
                         "behind the side records' `traffic`); the line's metric is BASELINE.json's only with `synthetic`")
    ap.add_argument("--unsorted", action="store_true",
                    help="do not length-bucket the synthetic corpus (BASELINE.json's configs are length-bucketed)")
    args = ap.parse_args()

    # N ranks, whichever way the script is started: under torch.distributed.run (the contract's N > 1 form: RANK / WORLD_SIZE
    # in the environment) --gpus must agree with it; started plainly with --gpus N > 1 the script re-executes itself under
    # torch.distributed.run with N processes -- a plain `python bench.py --gpus 8` must never run one rank and print n_gpus 1.
    if os.environ.get("WORLD_SIZE") is None and (args.gpus or 1) > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is None:
        args.gpus = world
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a number for another job size" % (args.gpus, world))
    if world > 1:       # the corpus generators of the ranks share the host's cores
        os.environ.setdefault("SPMX_SYNTH_WORKERS", str(max(1, (os.cpu_count() or 1) // world)))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.sentences is None:
        args.sentences = 12_500_000 if world > 1 else 10_000_000
    algos = ["all_gather", "p2p_exact"] if args.gather_algo == "both" else [args.gather_algo]
    # SPMX_BENCH_ONE_RANK_GATHER=1: the N > 1 control flow (process group over RCCL, both gather algorithms, reserved CUs, the
    # watchdog) with ONE rank -- the only way to run that code on a one-GPU box; the line it prints is not a scaling figure
    multi = world > 1 or os.environ.get("SPMX_BENCH_ONE_RANK_GATHER") == "1"
    gather_modes = ([] if not multi else ((["ids:" + a for a in algos] + ["none"]) if args.gather == "both"
                                           else (["ids:" + a for a in algos] if args.gather == "ids" else ["none"])))
    if multi and any(m.startswith("ids") for m in gather_modes):
        # an all-gather in flight needs CUs of its own: the persistent encode grids would otherwise hold every CU until
        # they end, and the gather of batch k would run after batch k + 1's encode instead of under it
        os.environ.setdefault("SPMX_RESERVE_CUS", "16")
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")

    import torch
    from sentencepiece_amd.processor import SentencePieceProcessor
    from sentencepiece_amd import sharding

    # Test seam (tests/test_bench_multirank.py): SPMX_BENCH_DRYRUN=1 runs THIS script's multi-rank control flow -- the
    # gather modes, the watchdog, the one JSON line -- on CPU tensors over gloo, with the emulated library injected by
    # the test's runner.  It measures nothing and is not a product path: libspmx itself has no CPU path.
    dry = os.environ.get("SPMX_BENCH_DRYRUN") == "1"
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    sync = (lambda: None) if dry else torch.cuda.synchronize
    # The corpus first: the generator forks a process pool, which is safest before this process has a HIP context
    # or an RCCL communicator.  Weak scaling: every rank draws its own shard of the generator (seed + rank).
    c5 = args.model.startswith("c5_")
    text, offs = corpus_for(args.model, args.sentences, 20250227 + rank, args.unsorted, args.corpus)
    if args.corpus != "synthetic":
        args.sentences = len(offs) - 1
        args.no_second_model = args.no_side_configs = True
    second = None
    if world == 1 and args.model == "uni32k" and not args.no_second_model and \
            os.path.exists(os.path.join(ROOT, "tests", "golden", "uni32k_w16.model")):
        second = ("uni32k_w16",) + corpus_for("uni32k_w16", args.sentences, 20250227, args.unsorted)
    if not dry:
        torch.cuda.set_device(local)
    dev = torch.device("cpu") if dry else torch.device("cuda", local)
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    blob = model_blob(args.model)
    sp = SentencePieceProcessor(model_proto=blob, device=local)

    n = len(offs) - 1
    d_text = torch.from_numpy(text).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_ids, d_io, total = sp.EncodeDevice(d_text, d_offs)          # sizes the output once
    d_ids = torch.empty(int(total) + int(total) // 16 + 64, dtype=torch.int32, device=dev)
    # ids travel as int16 when the vocabulary allows it (half the bytes on the point-to-point xGMI links)
    wire = torch.int16 if sp.GetPieceSize() <= 32768 else None
    # a sentence has at most one id per raw byte + 1 (+ the ids of the extra options): the host knows the longest sentence
    # of its shard from the offsets, so the per-sentence counts of the exact-size gather travel in one or two bytes
    max_count = (int(np.diff(offs.astype(np.int64)).max()) + 1 + 8) if n else 1

    def timed(run_step, wait=None):
        for _ in range(args.warmup):
            run_step()
        if wait:
            wait()
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tot = run_step()
        if wait:
            wait()
        sync()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, tot

    def encode_step():
        return sp.EncodeDevice(d_text, d_offs, d_ids, d_io)[2]

    def run_mode(mode):
        if not mode.startswith("ids:"):
            return timed(encode_step)
        g = sharding.IdGatherer(dist, dev, wire_dtype=wire, depth=GATHER_DEPTH, algo=mode[4:])
        g.reserve(d_ids.numel(), d_io.numel(), torch.int32, d_io.dtype)      # agreed once, before the loop

        def step_ids():
            tot = sp.EncodeDevice(d_text, d_offs, d_ids, d_io)[2]
            g(d_ids, tot, d_io, max_count=max_count)
            return tot
        r = timed(step_ids, g.wait)
        del g
        return r

    def gather_desc(mode):
        if not mode.startswith("ids"):
            return mode
        return ("%s (%s on the wire, per-sentence counts in %d byte(s) where exact sizes travel, capacities agreed once, "
                "%d gathers in flight, %s CUs left to RCCL)"
                % (mode, "int16" if wire is not None else "int32", sharding.IdGatherer.count_width(max_count), GATHER_DEPTH,
                   os.environ.get("SPMX_RESERVE_CUS", "0")))

    results = {}
    late_modes = []
    if not multi:
        results["n/a"] = timed(encode_step)
    else:
        # The no-gather form is timed first and makes the line (its only collectives are the barrier and the MAX of the
        # timing).  Every gather algorithm is timed AFTER the line is assembled, under a watchdog: none of them has run on
        # more than one GPU before the driver's node, and if one hangs, every rank gives up after the deadline and rank 0
        # still prints the line -- with the figures it has, `config.gather` saying which.  (Without "none" among the modes,
        # --gather ids, the first algorithm makes the line as before.)
        first_ids = next((m for m in gather_modes if m.startswith("ids:")), None)
        base_none = "none" in gather_modes
        for mode in gather_modes:
            if mode.startswith("ids:") and (base_none or mode != first_ids):
                late_modes.append(mode)
            else:
                results[mode] = run_mode(mode)
    with_ids = [m for m in results if m.startswith("ids:")]
    # the headline is WITH the gather (the north star): the faster of the algorithms timed in this run
    head = min(with_ids, key=lambda m: results[m][0]) if with_ids else ("none" if "none" in results else "n/a")
    dt, total = results[head]
    # a second, profiled loop for the per-kernel numbers (HIP events around every encode launch)
    sp.SetProfiling(True)
    prof = []
    for _ in range(args.steps):
        encode_step()
        prof.append(sp.LastProfile())
    sync()
    sp.SetProfiling(False)
    if dist is not None:
        tot_t = torch.tensor([float(len(text)), float(total)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot_t)
        job_bytes, job_ids = float(tot_t[0].item()), float(tot_t[1].item())
    else:
        job_bytes, job_ids = float(len(text)), float(total)

    if rank == 0:
        ms = dt / args.steps * 1e3
        # dominant kernel = the launch that does the most of the batch's work (algorithmic bytes).  Since round 4 the general
        # launch over the ~1 % of the sentences that are not plain ASCII runs NEXT TO the first word round on a second
        # stream: it may last longer than the word round it hides behind (it is bound by the latency of its longest
        # sentences), but it is not where the batch's bytes go; its own figures are in `roofline.beside`.
        ncls = len(prof[0]["classes"])
        k_ms = [sum(p["classes"][c]["kernel_ms"] for p in prof) / len(prof) for c in range(ncls)]
        k_bytes = [prof[-1]["classes"][c]["bytes"] if prof[-1]["classes"][c]["kernel"] else 0 for c in range(ncls)]
        dom = int(np.argmax(k_bytes))
        longest = int(np.argmax(k_ms))
        cls = prof[-1]["classes"][dom]
        achieved = cls["bytes"] / (k_ms[dom] * 1e-3) / 1e9 if k_ms[dom] > 0 else 0.0
        kname = cls["kernel"]
        sha = kernel_sources_sha()
        traffic, traffic_lower, traffic_note = traffic_on_record(args.model if args.corpus == "synthetic" else args.model + "@" + args.corpus, args.sentences, kname)
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
            "config": {"workload": ("side corpus `%s` (see the default line's record of that name), %s model, %d sentences, resident in HBM"
                                    % (args.corpus, args.model, n)) if args.corpus != "synthetic" else
                                   "configs[%d]: %s model, %d synthetic %s sentences per GPU, mean %.1f B, "
                                   "%s, resident in HBM"
                                   % (4 if c5 else ((1 if world == 1 else 3) if sp.model_type() == 1 else 2), args.model, n,
                                      "mixed-script power-law [16, 4096] B" if c5 else "ASCII", len(text) / n,
                                      "in generator order (not length-bucketed)" if args.unsorted else "length-bucketed"),
                       "model": args.model, "sentences_per_gpu": n, "ids_per_sentence": job_ids / (world * n),
                       "gather": gather_desc(head) if multi else "n/a",
                       "gather_bound": gather_bound(world if world > 1 else 8, n, job_ids / (world * n), 2 if wire is not None else 4,
                                                    sharding.IdGatherer.count_width(max_count),
                                                    (results["none"][0] if "none" in results else dt) / args.steps * 1e3),
                       "sharding": "dp%d by sentence" % world,
                       "handle": sp.HandleInfo(),      # device bytes of the model's tables, spmx_create's wall-clock ms
                       "timed_loop": "profiling off; roofline.* comes from a second loop of the same steps with HIP events on"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_uncorrected": traffic_lower,
                         "kernel_sources_sha": sha, "kernel": kname,
                         "kernel_ms": k_ms[dom], "algorithmic_bytes_per_launch": cls["bytes"],
                         "sentences_per_launch": cls["sentences"],
                         "all_kernels_ms": {prof[-1]["classes"][c]["kernel"]: round(k_ms[c], 4)
                                            for c in range(ncls) if prof[-1]["classes"][c]["kernel"]},
                         "beside": None if longest == dom else {
                             "kernel": prof[-1]["classes"][longest]["kernel"], "kernel_ms": k_ms[longest],
                             "sentences_per_launch": prof[-1]["classes"][longest]["sentences"],
                             "algorithmic_bytes_per_launch": prof[-1]["classes"][longest]["bytes"],
                             "what": "the launch with the longest duration of the step: it runs on the second stream next to "
                                     "the dominant kernel, over the sentences classify set aside"},
                         "pipeline": {"algorithmic_bytes": sum(k_bytes), "ms": ms,
                                      "achieved": sum(k_bytes) / (ms * 1e-3) / 1e9, "frac": sum(k_bytes) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "what": "every launch's algorithmic bytes over the whole step (classify, scan, compact included in the time)"},
                         "phase_cycles": cls.get("phase_cycles"), "path": prof[-1]["path"],
                         "pipeline_ms": sum(p["total_ms"] for p in prof) / len(prof)},
        }
        if world == 1 and out["roofline"]["beside"] is not None and not dry:
            # the dominant kernel's duration when nothing shares the CUs with it: one profiled step on a second handle that
            # runs the general launch BEFORE the word rounds (SPMX_NO_OVERLAP=1, read at load)
            try:
                os.environ["SPMX_NO_OVERLAP"] = "1"
                sp1 = SentencePieceProcessor(model_proto=blob, device=local)
                os.environ.pop("SPMX_NO_OVERLAP", None)
                sp1.SetProfiling(True)
                alone = []
                for _ in range(3):
                    sp1.EncodeDevice(d_text, d_offs, d_ids, d_io)
                    alone.append([c["kernel_ms"] for c in sp1.LastProfile()["classes"] if c["kernel"] == kname][0])
                sync()
                a_ms = sorted(alone)[1]
                out["roofline"]["alone"] = {"kernel_ms": a_ms, "achieved": cls["bytes"] / (a_ms * 1e-3) / 1e9,
                                            "frac": cls["bytes"] / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "what": "the same kernel with the general launch run before it instead of beside it"}
                del sp1
            except Exception as e:
                os.environ.pop("SPMX_NO_OVERLAP", None)
                out["roofline"]["alone"] = {"failed": repr(e)[:200]}
        if multi:
            for mode, (mdt, _) in results.items():
                key = mode.replace("ids:", "")
                out["value_gather_%s" % key] = world * n * args.steps / mdt
                out["ms_per_step_gather_%s" % key] = mdt / args.steps * 1e3
        io_h = None
        if second is not None:
            name, t2, o2 = second
            sp2 = SentencePieceProcessor(model_proto=model_blob(name), device=local)
            dt2_text = torch.from_numpy(t2).to(dev)
            dt2_offs = torch.from_numpy(o2.view(np.int64)).to(dev)
            i2, io2, tot2 = sp2.EncodeDevice(dt2_text, dt2_offs)
            i2 = torch.empty(int(tot2) + 64, dtype=torch.int32, device=dev)
            for _ in range(args.warmup):
                sp2.EncodeDevice(dt2_text, dt2_offs, i2, io2)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sp2.EncodeDevice(dt2_text, dt2_offs, i2, io2)
            sync()
            d2 = time.perf_counter() - t0
            sp2.SetProfiling(True)
            sp2.EncodeDevice(dt2_text, dt2_offs, i2, io2)
            p2 = sp2.LastProfile()
            out["long_piece_model"] = {"model": name, "value": (len(o2) - 1) * args.steps / d2, "unit": "sentences/s",
                                       "ms_per_step": d2 / args.steps * 1e3, "mean_bytes": len(t2) / (len(o2) - 1),
                                       "kernel": max(p2["classes"], key=lambda c: c["bytes"] if c["kernel"] else 0)["kernel"],
                                       "kernel_ms": max(p2["classes"], key=lambda c: c["bytes"] if c["kernel"] else 0)["kernel_ms"],
                                       "kernels_ms": {c["kernel"]: round(c["kernel_ms"], 4) for c in p2["classes"] if c["kernel"]},
                                       "vs_headline": ((len(o2) - 1) * args.steps / d2) / out["value"],
                                       "what": "the C2 recipe with words of up to 16 letters: longest piece 17 bytes, score ring of 18 entries"}
            dom2 = max((c for c in p2["classes"] if c["kernel"]), key=lambda c: c["bytes"], default=None)
            if dom2 is not None and dom2["kernel_ms"] > 0:
                ach2 = dom2["bytes"] / (dom2["kernel_ms"] * 1e-3) / 1e9
                out["long_piece_model"]["roofline"] = {"kernel": dom2["kernel"], "kernel_ms": dom2["kernel_ms"],
                                                       "sentences_per_launch": dom2["sentences"],
                                                       "algorithmic_bytes_per_launch": dom2["bytes"], "achieved": ach2,
                                                       "unit": "GB/s", "frac": ach2 / HBM_PEAK_GBS}
                out["long_piece_model"]["roofline"]["traffic"], _, out["long_piece_model"]["roofline"]["traffic_note"] = \
                    traffic_on_record(name, len(o2) - 1, dom2["kernel"])
            try:      # every sentence of this corpus against the compiled reference, as for the headline (round-4 verdict: the one record without it)
                io2_h = io2.cpu().numpy()
                out["long_piece_model"]["probe_ids_bit_exact"], out["long_piece_model"]["probe"] = probe_exact(
                    t2, o2, model_blob(name), i2[:int(io2_h[-1])].cpu().numpy(), io2_h)
            except Exception as e:
                out["long_piece_model"]["probe_ids_bit_exact"] = None
                out["long_piece_model"]["probe_error"] = repr(e)[:200]
            del sp2, dt2_text, dt2_offs, i2, io2
        if world == 1 and args.model == "uni32k" and not args.no_side_configs:
            # the other single-GPU configurations of BASELINE.json, compact: c3 (the same corpus through the 32k BPE
            # model), c5 (250k-piece unigram, mixed script, 1 M sentences), documents (8192 x 16 KB, 256 x 1 MiB)
            try:
                out["c3"] = side_bench(SentencePieceProcessor, torch, dev, "bpe32k", model_blob("bpe32k"), text, offs,
                                       args.steps, args.warmup, "configs[2]: 32k BPE, the same %d sentences" % n)
            except Exception as e:
                out["c3"] = {"failed": repr(e)[:300]}
            # a Llama-style BPE model (byte fallback, extra whitespace kept, pieces of space symbols only, 1000 pieces: words
            # split finely): the word kernels with the call-local memo's wide entries (round 3: 121 M through the stream kernel)
            try:
                out["llama_style_bpe"] = side_bench(SentencePieceProcessor, torch, dev, "bpe1k_llama", model_blob("bpe1k_llama"), text, offs,
                                                    max(1, args.steps // 2), 1, "bpe1k_llama (tests/golden), the same %d sentences" % n)
            except Exception as e:
                out["llama_style_bpe"] = {"failed": repr(e)[:300]}
            # text the word memo does not fit by construction: the C2 recipe with 5 % of its tokens fresh random words, and
            # the novel of the reference's own tests x 2000 (8.6 M lines), each with every sentence against the reference
            try:
                from sentencepiece_amd import synth
                for key, mk, what in (("natural_open_vocab", lambda: synth.open_vocab_corpus(n, seed=20250301),
                                       "the C2 recipe, 5 %% of the word draws replaced by fresh random words (200 k of them), %d sentences" % n),
                                      ("natural_botchan_x2000", lambda: synth.repeated_file_corpus(os.path.join(ROOT, "tests", "golden", "botchan.txt"), 2000),
                                       "tests/golden/botchan.txt x 2000, lines shuffled per repetition, in file order (not length-bucketed)")):
                    tn, on = mk()
                    out[key] = side_bench(SentencePieceProcessor, torch, dev, "uni32k", blob, tn, on, args.steps, args.warmup, what,
                                          corpus=key.split("_", 1)[1].replace("_x2000", ""))
                    out[key]["vs_headline"] = out[key]["value"] / out["value"]
                    del tn, on
            except Exception as e:
                out["natural"] = {"failed": repr(e)[:300]}
            try:
                t5, o5 = corpus_for("c5_250k", 1_000_000, 20250227, False)
                out["c5"] = side_bench(SentencePieceProcessor, torch, dev, "c5_250k", model_blob("c5_250k"), t5, o5,
                                       args.steps, args.warmup,
                                       "configs[4]: 250k-piece unigram, 1 M mixed-script sentences, power-law [16, 4096] characters")
                # ... and the same model with byte_fallback on ("both off and on")
                out["c5_bf"] = side_bench(SentencePieceProcessor, torch, dev, "c5_250k_bf", model_blob("c5_250k_bf"), t5, o5,
                                          args.steps, args.warmup, "configs[4] with byte_fallback: the same 1 M sentences")
                del t5, o5
            except Exception as e:
                out.setdefault("c5", {"failed": repr(e)[:300]})
                out.setdefault("c5_bf", {"failed": repr(e)[:300]})
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import docs_rate
                for key, nd, nb in (("docs_16k", 8192, 16384), ("docs_1m", 256, 1 << 20)):
                    td, od = docs_rate.make_docs(nd, nb)
                    out[key] = side_bench(SentencePieceProcessor, torch, dev, "uni32k", blob, td, od, max(1, args.steps // 2), 1,
                                          "%d documents of ~%d bytes (C2 sentences joined by spaces), uni32k" % (nd, nb), corpus=key)
                    out[key]["mb_per_s"] = len(td) / 1e6 / (out[key]["ms_per_step"] * 1e-3)
                    del td, od
            except Exception as e:
                out["docs"] = {"failed": repr(e)[:300]}
        if world == 1 and not args.no_side_configs:
            # host arrays in, host arrays out (SURVEY 8d's end-to-end figure): pack is the caller's, the call copies the text
            # to the GPU in chunks, encodes, copies the ids back -- the chunk pipeline of spmx_encode_batch (PCIe-inclusive;
            # never `value`)
            try:
                sp.SetProfiling(False)
                import ctypes as C
                lib = sp._lib
                runs = []
                for _ in range(9):               # the C call itself: the arrays it returns belong to the library (pinned, recycled)
                    p_ids, p_off = C.c_void_p(), C.c_void_p()
                    t0 = time.perf_counter()
                    rc = lib.spmx_encode_batch(sp._h, text.ctypes.data, offs.ctypes.data, n, C.byref(p_ids), C.byref(p_off))
                    runs.append(time.perf_counter() - t0)
                    assert rc == 0, lib.spmx_last_error(None)
                    h_io = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
                    h_ids = np.ctypeslib.as_array(C.cast(p_ids, C.POINTER(C.c_int32)), shape=(int(h_io[-1]),)).copy() if len(runs) == 9 else None
                    lib.spmx_free(p_ids)
                    lib.spmx_free(p_off)
                # The first call sizes the library's pools -- pinned staging and a device workspace per worker, each for the
                # batch's LARGEST chunk (api.cc Workspace::reserve_text_bytes): ~400 ms, once per handle -- and is reported,
                # not timed (scripts/host_calls_probe.py prints a handle's calls in order: 47 - 57 ms from the second on; before
                # the workspaces were sized for the largest chunk the second and third call still grew them: 115 and 60 ms).
                in_order = list(runs)
                runs = sorted(runs[1:])
                out["end_to_end"] = {"what": "spmx_encode_batch: packed text + offsets in host memory -> ids + offsets in host memory "
                                             "(H2D, kernels and D2H of successive chunks overlapped), %d sentences" % n,
                                     "value": n / runs[len(runs) // 2], "best": n / runs[0], "unit": "sentences/s",
                                     "seconds": runs, "warmup_calls": 1, "seconds_in_call_order": in_order, "gb_text_per_s": len(text) / runs[len(runs) // 2] / 1e9,
                                     # (offsets AND ids of the last call against the device-resident run's)
                                     "ids_equal_device_run": bool(np.array_equal(np.asarray(h_io).astype(np.int64), d_io.cpu().numpy()) and
                                                                  h_ids is not None and
                                                                  np.array_equal(h_ids, d_ids[:int(h_io[-1])].cpu().numpy()))}
                del h_io, h_ids
            except Exception as e:
                out["end_to_end"] = {"failed": repr(e)[:300]}
            # The reference's ONLY batch entry is its Python wrapper (SURVEY finding 1): sp.encode(list[str]) -> list[list[int]].
            # The same call through this engine's Python mirror, and -- for scale -- the pip wheel of the reference on the
            # same list with its own thread pool (python/src/sentencepiece/sentencepiece.i:210-267): both pay CPython for
            # every list and int they hand back, which bounds either at a few M sentences/s whatever encodes.
            try:
                k = min(n, 2_000_000)
                pick = np.linspace(0, n - 1, num=k).astype(np.int64)
                raw = text.tobytes()
                o64 = offs.astype(np.int64)
                strs = [raw[o64[i]:o64[i + 1]].decode("utf-8", "replace") for i in pick]
                sp.encode(strs[:1000])
                t0 = time.perf_counter()
                got = sp.encode(strs)
                dt_py = time.perf_counter() - t0
                rec = {"what": "sp.encode(list of %d str) -> list of lists of int through sentencepiece_amd.processor "
                               "(views handed to spmx_encode_batch_views, lists built by the C extension)" % k,
                       "value": k / dt_py, "unit": "sentences/s", "seconds": dt_py}
                try:
                    import sentencepiece as ref_wheel
                    rw = ref_wheel.SentencePieceProcessor(model_proto=blob)
                    threads = min(os.cpu_count() or 1, 64)
                    rw.encode(strs[:1000], num_threads=threads)
                    t0 = time.perf_counter()
                    want = rw.encode(strs, num_threads=threads)
                    dt_w = time.perf_counter() - t0
                    rec["reference_wheel"] = {"version": ref_wheel.__version__, "value": k / dt_w, "seconds": dt_w, "threads": threads,
                                              "what": "the pip wheel's sp.encode(list, num_threads) on the same list (its ids are "
                                                      "not this engine's parity bar -- the compiled reference is -- only its rate is read)",
                                              "lists_equal": bool(want == got)}
                    rec["vs_reference_wheel"] = dt_w / dt_py
                    del want
                except Exception as e:
                    rec["reference_wheel"] = {"failed": repr(e)[:200]}
                out["end_to_end_python"] = rec
                del strs, got, raw
            except Exception as e:
                out["end_to_end_python"] = {"failed": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            io_h = d_io.cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(text, offs, blob, np.diff(io_h), d_ids[:int(io_h[-1])].cpu().numpy(), io_h)
    else:
        out = None
    with_gather = head.startswith("ids")       # the line's `value` is from a run WITH the gather (the north star's form)
    for mode in late_modes:
        import threading
        key = mode.replace("ids:", "")
        deadline = max(float(os.environ.get("SPMX_BENCH_GATHER_DEADLINE_S", "120")), 40.0 * dt)

        def give_up(key=key, deadline=deadline, with_gather=with_gather):
            if rank == 0:
                out["gather_%s" % key] = "no result within %.0f s: given up, the line is from the other modes" % deadline
                if not with_gather:
                    out["config"]["gather"] = "none (no gather algorithm completed: `value` is the encode without the gather)"
                print(json.dumps(out), flush=True)
            os._exit(0)
        dog = threading.Timer(deadline, give_up)
        dog.daemon = True
        dog.start()
        mdt, _ = run_mode(mode)
        dog.cancel()
        better = (not with_gather) or mdt < dt  # the first gather that completes makes the headline; a faster one replaces it
        if rank == 0:
            out["value_gather_%s" % key] = world * n * args.steps / mdt
            out["ms_per_step_gather_%s" % key] = mdt / args.steps * 1e3
            if better:
                out["value"] = world * n * args.steps / mdt
                out["ms_per_step"] = mdt / args.steps * 1e3
                out["gb_text_per_s"] = job_bytes * args.steps / mdt / 1e9
                out["config"]["gather"] = gather_desc(mode)
        if better:
            dt, head, with_gather = mdt, mode, True      # (mdt is the max over ranks: every rank takes the same branch)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
