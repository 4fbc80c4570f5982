"""Multi-GPU form of the batch encode: one process per GPU, sentences sharded by
contiguous blocks of equal byte count, one variable-length all-gather of the id
streams (+ per-sentence id offsets) per batch over RCCL / xGMI.

Sentences are independent (the reference's only batch form is a bag of
per-sentence Encode calls, python/src/sentencepiece/sentencepiece.i:245-267),
so the data path needs no collective; the gather exists because the north star
asks for the complete id output on every rank.  It runs as an async RCCL
operation so that the next batch's kernels overlap it.

Works on CUDA tensors with the "nccl" backend (= RCCL on ROCm) and on CPU
tensors with "gloo" (the world_size-2 tests).
"""
import numpy as np
import torch


def shard_bounds(offsets, world):
    """Sentence index bounds [world + 1] of contiguous shards with ~equal bytes.

    ``offsets`` is the packed buffer's uint64/int64 [n + 1] array (numpy)."""
    offs = np.asarray(offsets).astype(np.int64)
    n = len(offs) - 1
    base, total = int(offs[0]), int(offs[n] - offs[0])
    targets = base + (np.arange(1, world, dtype=np.float64) * (total / world)).astype(np.int64)
    cuts = np.searchsorted(offs, targets, side="left")
    b = np.concatenate([[0], np.clip(cuts, 0, n), [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


class IdGatherer:
    """All-gatherv of (ids[:total], id_offsets) from every rank.

    Call ``g(ids, total, id_offsets)`` after each encode: the collective is
    issued asynchronously; ``wait()`` blocks the current stream (and, on CPU,
    the host) until the last one has finished.  ``result()`` returns the
    per-rank ``(ids, id_offsets)`` views of the last gather."""

    def __init__(self, dist, device, group=None, wire_dtype=None):
        """``wire_dtype``: dtype the ids travel in (default: as given).  xGMI is point-to-point, so a ring
        all-gather is bound by one link; a vocabulary below 32768 fits ``torch.int16`` and halves the payload.
        ``result()`` widens back to the dtype of the ids passed in."""
        self.dist, self.device, self.group = dist, device, group
        self.wire = wire_dtype
        self._dtype = None
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._work = []
        self._cap = self._ocap = 0
        self._out = self._oout = self._tot = None
        self._pad = self._opad = None

    def _all_gather(self, out, inp):
        if out.dtype == torch.int16:      # no 16-bit integer type in NCCL / gloo: an all-gather only moves bytes
            out, inp = out.view(torch.uint8), inp.view(torch.uint8)
        if self.dist.get_backend(self.group) == "nccl":
            return self.dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)
        return self.dist.all_gather(list(out.view(self.world, -1).unbind(0)), inp, group=self.group, async_op=True)

    def _agree(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def __call__(self, ids, total, id_offsets=None):
        self.wait()
        # capacities are agreed once and only grow (a MAX all-reduce, off the steady-state path)
        need = self._agree(max(int(total), 1))
        self._dtype = ids.dtype
        wire = self.wire or ids.dtype
        if need > self._cap:
            self._cap = need + need // 8
            self._out = torch.empty(self.world * self._cap, dtype=wire, device=self.device)
            self._pad = torch.empty(self._cap, dtype=wire, device=self.device)
        # staged copy: ranks hold different totals (padding), and the caller's
        # buffer is free for the next batch while the collective is in flight
        self._pad[:int(total)].copy_(ids[:int(total)])
        src = self._pad
        self._tot_in = torch.tensor([int(total)], dtype=torch.int64, device=self.device)
        self._tot = torch.empty(self.world, dtype=torch.int64, device=self.device)
        self._work = [self._all_gather(self._tot, self._tot_in), self._all_gather(self._out, src[:self._cap])]
        if id_offsets is not None:
            oneed = self._agree(id_offsets.numel())
            if oneed > self._ocap:
                self._ocap = oneed
                self._oout = torch.empty(self.world * oneed, dtype=id_offsets.dtype, device=self.device)
                self._opad = torch.zeros(oneed, dtype=id_offsets.dtype, device=self.device)
            self._opad[:id_offsets.numel()].copy_(id_offsets)
            osrc = self._opad
            self._n_in = torch.tensor([id_offsets.numel() - 1], dtype=torch.int64, device=self.device)
            self._n = torch.empty(self.world, dtype=torch.int64, device=self.device)
            self._work += [self._all_gather(self._n, self._n_in), self._all_gather(self._oout, osrc)]
        else:
            self._n = None

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = []

    def result(self):
        self.wait()
        tot = self._tot.cpu().tolist()
        ids = [self._out[r * self._cap: r * self._cap + tot[r]].to(self._dtype) for r in range(self.world)]
        offs = None
        if self._n is not None:
            ns = self._n.cpu().tolist()
            offs = [self._oout[r * self._ocap: r * self._ocap + ns[r] + 1] for r in range(self.world)]
        return ids, offs


def encode_sharded(encode_fn, text, offsets, dist, device, group=None):
    """Every rank holds the same packed batch; rank r encodes shard r and all
    ranks end up with the full CSR ``(ids, id_offsets)`` in the original order.

    ``encode_fn(text_shard uint8 tensor, offsets_shard int64 tensor) ->
    (ids tensor, id_offsets int64 tensor, total)`` is the single-GPU encode
    (``SentencePieceProcessor.EncodeDevice``)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    offs_np = offsets.cpu().numpy() if isinstance(offsets, torch.Tensor) else np.asarray(offsets)
    b = shard_bounds(offs_np, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    o = torch.as_tensor(offs_np[lo:hi + 1].astype(np.int64) - int(offs_np[lo]), device=device)
    t = torch.as_tensor(text, device=device)[int(offs_np[lo]):int(offs_np[hi])]
    ids, io, total = encode_fn(t, o)
    g = IdGatherer(dist, device, group)
    g(ids, total, io)
    parts, offs = g.result()
    full_ids = torch.cat(parts)
    bases = np.concatenate([[0], np.cumsum([p.numel() for p in parts])])
    full_off = torch.cat([offs[r][:-1] + int(bases[r]) for r in range(world)] +
                         [torch.tensor([int(bases[-1])], dtype=offs[0].dtype, device=device)])
    return full_ids, full_off
