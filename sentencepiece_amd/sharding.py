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


class _Works:
    """Several work handles behind one wait()."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()


class IdGatherer:
    """All-gatherv of (ids[:total], id_offsets) from every rank, asynchronous and free of host synchronisation in
    the steady state.

    ``reserve(ids_capacity, offsets_capacity)`` -- collective, ONCE (and again only if a rank's buffers grow): the ranks
    agree on the padded sizes by one MAX all-reduce.  A rank passes what its output buffers hold: an encode can never
    produce more ids than fit its output buffer, so no later batch can outgrow the agreement.
    ``g(ids, total, id_offsets)`` after each encode: a staged copy on the current stream, then the collective on the
    process group's own stream (``async_op``); ``depth`` gathers may be in flight (double buffering), so batch k's
    gather overlaps the kernels of batches k + 1 and k + 2.  Nothing here reads a device value on the host.
    ``wait()`` makes the current stream wait for everything in flight; ``result()`` returns the per-rank
    ``(ids, id_offsets)`` views of the last gather (it is the consumer that synchronises, not the gatherer)."""

    def __init__(self, dist, device, group=None, wire_dtype=None, depth=2, algo="all_gather"):
        """``wire_dtype``: dtype the ids travel in (default: as given).  xGMI is point-to-point, so a ring
        all-gather is bound by one link; a vocabulary below 32768 fits ``torch.int16`` and halves the payload.
        ``result()`` widens back to the dtype of the ids passed in.
        ``algo``: "all_gather" (the library's collective) or "p2p": world - 1 sends of the rank's own buffer and
        world - 1 receives per gather, posted as one batch (``batch_isend_irecv``) -- on a fully connected xGMI node
        every peer is one hop away, so the sends go out on all links at once instead of round a ring."""
        self.dist, self.device, self.group = dist, device, group
        self.wire = wire_dtype
        self.algo = algo
        # "p2p_exact": every rank sends exactly its `total` ids (and its per-sentence id COUNTS as int32, not n + 1
        # 64-bit offsets) straight to every peer -- no padding to the largest rank's capacity.  The receivers must know the
        # sizes before they post their receives: two integers per rank, exchanged host-side over a gloo group of its
        # own (the host knows its own total: EncodeDevice returns it), which does not queue behind the gathers in flight --
        # ASYNCHRONOUSLY: the sizes of batch k travel while batch k + 1 is encoded, and batch k's transfers are posted at
        # the next call (_call_exact / _post_exact), so no call waits for a host round trip.
        self._cpu_group = None
        if algo == "p2p_exact" and dist.get_backend(group) != "gloo":
            # (construction is collective over the ranks of `group`: the side channel spans exactly those)
            self._cpu_group = dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None,
                                             backend="gloo")
        self._dtype = None
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.depth = max(1, int(depth))
        self._cap = self._ocap = 0
        self._slots = []          # per slot: dict(pad, out, opad, oout, tot_in, tot, n_in, n, work)
        self._k = 0               # gathers issued
        self._last = None
        self._pending = None      # (p2p_exact) the slot whose batch is staged and not yet posted

    def _all_gather(self, out, inp):
        if out.dtype == torch.int16:      # no 16-bit integer type in NCCL / gloo: an all-gather only moves bytes
            out, inp = out.view(torch.uint8), inp.view(torch.uint8)
        if self.algo == "p2p" and self.world > 1:
            rows = out.view(self.world, -1)
            rows[self.rank].copy_(inp)
            ops = []
            for k in range(1, self.world):           # peer order staggered by rank: no two ranks start on the same peer
                to, frm = (self.rank + k) % self.world, (self.rank - k) % self.world
                ops.append(self.dist.P2POp(self.dist.isend, inp, self._peer(to), self.group))
                ops.append(self.dist.P2POp(self.dist.irecv, rows[frm], self._peer(frm), self.group))
            return _Works(self.dist.batch_isend_irecv(ops))
        if self.dist.get_backend(self.group) == "nccl":
            return self.dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)
        return self.dist.all_gather(list(out.view(self.world, -1).unbind(0)), inp, group=self.group, async_op=True)

    def _peer(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def reserve(self, ids_capacity, offsets_capacity=0, ids_dtype=torch.int32, offsets_dtype=torch.int64):
        """Collective.  Agrees the padded per-rank sizes (MAX over the ranks) and allocates the slots."""
        if self.algo == "p2p_exact" and self.world > 1:
            return                       # (exact sizes travel with every batch: nothing to agree up front; one rank alone takes the padded form)
        self.wait()
        t = torch.tensor([int(ids_capacity), int(offsets_capacity)], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        cap, ocap = (int(v) for v in t.cpu().tolist())
        if cap <= self._cap and ocap <= self._ocap and self._slots:
            return
        self._cap, self._ocap = max(cap, self._cap, 1), max(ocap, self._ocap)
        wire = self.wire or ids_dtype
        self._slots = []
        for _ in range(self.depth):
            sl = dict(pad=torch.zeros(self._cap, dtype=wire, device=self.device),
                      out=torch.empty(self.world * self._cap, dtype=wire, device=self.device),
                      tot_in=torch.zeros(1, dtype=torch.int64, device=self.device),
                      tot=torch.zeros(self.world, dtype=torch.int64, device=self.device),
                      n_in=torch.zeros(1, dtype=torch.int64, device=self.device),
                      n=torch.zeros(self.world, dtype=torch.int64, device=self.device), work=[], has_offs=False)
            if self._ocap:
                sl["opad"] = torch.zeros(self._ocap, dtype=offsets_dtype, device=self.device)
                sl["oout"] = torch.empty(self.world * self._ocap, dtype=offsets_dtype, device=self.device)
            self._slots.append(sl)

    @staticmethod
    def count_width(max_count):
        """Bytes a per-sentence id COUNT travels in (p2p_exact) given the caller's bound on ids per sentence -- e.g. the
        longest raw sentence of the shard + the extra ids, which the host knows from the input offsets without reading
        anything back from the device: 1 (uint8) up to 255, 2 (int16 pattern) up to 65535, else 4.  None: 4."""
        if max_count is None:
            return 4
        return 1 if max_count <= 255 else (2 if max_count <= 65535 else 4)

    def _call_exact(self, ids, total, id_offsets, max_count=None):
        """Two steps per batch, one call apart.  STAGE (now): the rank's ids and per-sentence counts are copied to the
        slot's send buffers and the two integers every receiver needs -- {total, sentences} of every rank -- start
        travelling over the side channel ASYNCHRONOUSLY.  POST (at the next call, or at wait() / result()): the sizes of
        the batch staged one call ago have long arrived (a whole encode lies in between), so reading them costs no round
        trip; the exact-size sends and receives of that batch are posted as one batch.  The host-side exchange is thus
        off the critical path of every batch (round 3 waited for it inside the call)."""
        dist = self.dist
        n = id_offsets.numel() - 1 if id_offsets is not None else 0
        wire = self.wire or ids.dtype
        self._dtype = ids.dtype
        if not self._slots or len(self._slots) != self.depth or "xout" not in self._slots[0]:
            self._slots = [dict(work=[], xout=None, xcnt=None, xsend=None, xcsend=None, staged=None) for _ in range(self.depth)]
        sl = self._slots[self._k % self.depth]
        if sl.get("staged") is not None:          # (depth 1: the slot's own batch is still waiting to be posted)
            self._post_exact(sl)
        for w in sl["work"]:
            w.wait()
        sl["work"] = []
        if sl["xsend"] is None or sl["xsend"].numel() < total or sl["xsend"].dtype != wire:
            sl["xsend"] = torch.empty(max(total + total // 8, 1), dtype=wire, device=self.device)
        sl["xsend"][:total].copy_(ids[:total])
        csend = None
        width = self.count_width(max_count)
        if id_offsets is not None:
            c64 = id_offsets[1:] - id_offsets[:-1]
            # (the width is the SENDER's promise; it travels with the sizes, so ranks may differ)
            csend = c64.to(torch.uint8) if width == 1 else (c64.to(torch.int32).to(torch.int16) if width == 2 else c64.to(torch.int32))
        sl["xcsend"] = csend
        meta = torch.tensor([total, n, width], dtype=torch.int64)
        metas = [torch.zeros(3, dtype=torch.int64) for _ in range(self.world)]
        mw = dist.all_gather(metas, meta, group=self._cpu_group if self._cpu_group is not None else self.group, async_op=True)
        sl["staged"] = (mw, metas, meta, total, n, id_offsets is not None, id_offsets.dtype if id_offsets is not None else None)
        prev = self._pending
        self._pending = sl
        if prev is not None and prev is not sl and prev.get("staged") is not None:
            self._post_exact(prev)                # the batch staged one call ago
        self._last = sl
        self._k += 1

    def _post_exact(self, sl):
        dist, world, rank = self.dist, self.world, self.rank
        mw, metas, _meta, total, n, has_offs, odtype = sl["staged"]
        sl["staged"] = None
        mw.wait()
        tots = [int(m[0]) for m in metas]
        ns = [int(m[1]) for m in metas]
        widths = [int(m[2]) for m in metas]
        wire = sl["xsend"].dtype
        sum_t, sum_n = sum(tots), sum(ns)
        if sl["xout"] is None or sl["xout"].numel() < sum_t or sl["xout"].dtype != wire:
            sl["xout"] = torch.empty(max(sum_t + sum_t // 8, 1), dtype=wire, device=self.device)
        t_base = np.concatenate([[0], np.cumsum(tots)]).astype(np.int64)
        n_base = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
        send = sl["xsend"][:total]
        sl["xout"][int(t_base[rank]):int(t_base[rank + 1])].copy_(send)
        csend = sl["xcsend"] if has_offs else None
        cb = None                                  # byte offsets of the ranks' count arrays in the receive buffer
        if has_offs:
            cb = np.concatenate([[0], np.cumsum([((ns[r] * widths[r] + 15) // 16) * 16 for r in range(world)])]).astype(np.int64)
            if sl["xcnt"] is None or sl["xcnt"].numel() < int(cb[-1]):
                sl["xcnt"] = torch.empty(max(int(cb[-1]) + int(cb[-1]) // 8, 16), dtype=torch.uint8, device=self.device)
            sl["xcnt"][int(cb[rank]):int(cb[rank]) + n * widths[rank]].copy_(csend.view(torch.uint8))

        def as_bytes(t):
            return t.view(torch.uint8) if t.dtype == torch.int16 else t
        ops = []
        for k in range(1, world):
            to, frm = (rank + k) % world, (rank - k) % world
            if total:
                ops.append(dist.P2POp(dist.isend, as_bytes(send), self._peer(to), self.group))
            if tots[frm]:
                ops.append(dist.P2POp(dist.irecv, as_bytes(sl["xout"][int(t_base[frm]):int(t_base[frm + 1])]), self._peer(frm), self.group))
            if csend is not None:
                if n:
                    ops.append(dist.P2POp(dist.isend, csend.view(torch.uint8), self._peer(to), self.group))
                if ns[frm]:
                    ops.append(dist.P2POp(dist.irecv, sl["xcnt"][int(cb[frm]):int(cb[frm]) + ns[frm] * widths[frm]], self._peer(frm), self.group))
        sl["work"] = [_Works(dist.batch_isend_irecv(ops))] if ops else []
        sl["exact"] = (tots, ns, t_base, n_base, has_offs, odtype, widths, cb)

    def __call__(self, ids, total, id_offsets=None, max_count=None):
        """``max_count`` (p2p_exact): the caller's bound on the ids of one sentence of this batch (count_width): the
        counts then travel in one or two bytes instead of four.  A bound that does not hold corrupts the counts."""
        total = int(total)
        if self.algo == "p2p_exact" and self.world > 1:
            return self._call_exact(ids, total, id_offsets, max_count)
        need_o = id_offsets.numel() if id_offsets is not None else 0
        if not self._slots or total > self._cap or need_o > self._ocap:
            # first use (or a caller that grew its buffers without reserve()): agree now -- collective, so every rank
            # must be in the same situation; reserve() up front keeps this off the steady-state path
            self.reserve(max(total, self._cap), max(need_o, self._ocap), ids.dtype,
                         id_offsets.dtype if id_offsets is not None else torch.int64)
        self._dtype = ids.dtype
        sl = self._slots[self._k % self.depth]
        for w in sl["work"]:          # the gather issued `depth` calls ago
            w.wait()
        # staged copy: ranks hold different totals (padding), and the caller's buffer is free for the next batch while
        # the collective is in flight
        sl["pad"][:total].copy_(ids[:total])
        sl["tot_in"].fill_(total)
        sl["work"] = [self._all_gather(sl["tot"], sl["tot_in"]), self._all_gather(sl["out"], sl["pad"])]
        sl["has_offs"] = id_offsets is not None
        if id_offsets is not None:
            sl["opad"][:need_o].copy_(id_offsets)
            sl["n_in"].fill_(need_o - 1)
            sl["work"] += [self._all_gather(sl["n"], sl["n_in"]), self._all_gather(sl["oout"], sl["opad"])]
        self._last = sl
        self._k += 1

    def _flush_exact(self):
        for sl in self._slots:
            if isinstance(sl, dict) and sl.get("staged") is not None:
                self._post_exact(sl)
        self._pending = None

    def wait(self):
        if self.algo == "p2p_exact" and self.world > 1:
            self._flush_exact()
        for sl in self._slots:
            for w in sl["work"]:
                w.wait()
            sl["work"] = []

    def result(self):
        if self.algo == "p2p_exact" and self.world > 1:
            self._flush_exact()
        sl = self._last
        for w in sl["work"]:
            w.wait()
        sl["work"] = []
        if "exact" in sl and sl.get("exact") is not None and self.algo == "p2p_exact" and self.world > 1:
            tots, ns, t_base, n_base, has_offs, odt, widths, cb = sl["exact"]
            ids = [sl["xout"][int(t_base[r]):int(t_base[r + 1])].to(self._dtype) for r in range(self.world)]
            offs = None
            if has_offs:
                offs = []
                for r in range(self.world):
                    raw = sl["xcnt"][int(cb[r]):int(cb[r]) + ns[r] * widths[r]]
                    if widths[r] == 1:
                        c = raw.to(odt)
                    elif widths[r] == 2:
                        c = raw.view(torch.int16).to(torch.int32).bitwise_and(0xFFFF).to(odt)
                    else:
                        c = raw.view(torch.int32).to(odt)
                    offs.append(torch.cat([torch.zeros(1, dtype=odt, device=c.device), torch.cumsum(c, 0)]))
            return ids, offs
        tot = sl["tot"].cpu().tolist()
        ids = [sl["out"][r * self._cap: r * self._cap + tot[r]].to(self._dtype) for r in range(self.world)]
        offs = None
        if sl["has_offs"]:
            ns = sl["n"].cpu().tolist()
            offs = [sl["oout"][r * self._ocap: r * self._ocap + ns[r] + 1] for r in range(self.world)]
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
    # the shard as a tensor of its own (a slice would hand the kernels a pointer at an arbitrary offset; they cope --
    # their loads are aligned on the absolute address -- but a fresh tensor is also what a real caller has)
    t = torch.as_tensor(text, device=device)[int(offs_np[lo]):int(offs_np[hi])].clone()
    ids, io, total = encode_fn(t, o)
    g = IdGatherer(dist, device, group)
    g(ids, total, io)
    parts, offs = g.result()
    full_ids = torch.cat(parts)
    bases = np.concatenate([[0], np.cumsum([p.numel() for p in parts])])
    full_off = torch.cat([offs[r][:-1] + int(bases[r]) for r in range(world)] +
                         [torch.tensor([int(bases[-1])], dtype=offs[0].dtype, device=device)])
    return full_ids, full_off
