"""Whole-corpus comparison of a device encode with the compiled reference (oracle/_ref), or with the oracle where the
reference is not built: EVERY sentence of the batch, ids one by one -- not a sample.  The reference runs its Encode loop
on all host cores over chunks of the corpus (about 2 M sentences/s on the GPU box: 10 M sentences in ~5 s).
TEST INFRASTRUCTURE ONLY (tests/, bench.py's probe): the product never imports this."""
import os
import time

import numpy as np


def checker(model_blob):
    """-> (encode_batch(text, offs) -> (ids, id_offsets), kind)"""
    from tests import refshim
    if refshim.available():
        h = refshim.RefLib().load(model_blob)
        threads = max(1, min(os.cpu_count() or 1, 64))
        return (lambda t, o: h.encode_batch(t, o, threads=threads)), "compiled reference (oracle/_ref), %d threads" % threads
    from tests import oraclelib
    h = oraclelib.OracleLib().load(model_blob)
    return h.encode_batch, "oracle (plain-C restatement), one thread"


def compare_all(text, offs, ids, id_offsets, model_blob, chunk=500_000, limit_seconds=None):
    """ids / id_offsets: the device's CSR for the packed batch (text, offs), on the host.
    -> dict(sentences, compared, differing, first (index or None), seconds, kind).  `limit_seconds`: stop starting new
    chunks after this long (compared < sentences then; the caller reports it)."""
    enc, kind = checker(model_blob)
    offs_i = np.asarray(offs).astype(np.int64)
    io = np.asarray(id_offsets).astype(np.int64)
    ids = np.asarray(ids)
    n = len(offs_i) - 1
    differing, first, compared = 0, None, 0
    t0 = time.perf_counter()
    for a in range(0, n, chunk):
        if limit_seconds is not None and time.perf_counter() - t0 > limit_seconds:
            break
        b = min(n, a + chunk)
        t = text[offs_i[a]:offs_i[b]]
        o = (offs_i[a:b + 1] - offs_i[a]).astype(np.uint64)
        rids, rio = enc(t, o)
        rio = np.asarray(rio).astype(np.int64)
        mine_io = io[a:b + 1] - io[a]
        mine = ids[io[a]:io[b]]
        compared += b - a
        if np.array_equal(mine_io, rio) and np.array_equal(mine, np.asarray(rids)):
            continue
        # some sentence of the chunk differs: count them
        same_len = np.diff(mine_io) == np.diff(rio)
        bad = ~same_len
        idx = np.flatnonzero(same_len)
        if len(idx):
            # sentences of equal length: compare their id ranges element-wise, reduce per sentence
            lens = np.diff(rio)[idx]
            src_m = np.repeat(mine_io[:-1][idx], lens) + _ragged(lens)
            src_r = np.repeat(rio[:-1][idx], lens) + _ragged(lens)
            ne = mine[src_m] != np.asarray(rids)[src_r]
            owner = np.repeat(np.arange(len(idx)), lens)
            bad_eq = np.zeros(len(idx), dtype=bool)
            np.logical_or.at(bad_eq, owner[ne], True)
            bad[idx[bad_eq]] = True
        k = np.flatnonzero(bad)
        differing += len(k)
        if first is None and len(k):
            first = int(a + k[0])
    return {"sentences": n, "compared": compared, "differing": int(differing), "first": first,
            "seconds": time.perf_counter() - t0, "kind": kind}


def _ragged(lens):
    total = int(lens.sum())
    starts = np.cumsum(lens) - lens
    return np.arange(total, dtype=np.int64) - np.repeat(starts, lens)
