"""ctypes helpers shared by the oracle and reference bindings: the pieces form (ids + spans + piece strings) and the
batch Normalize with alignment.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np


def encode_pieces(fn, h, text, offs):
    """-> (ids, begin, end, id_offsets, piece_blob bytes, piece_offsets uint64[total + 1])"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    cap = int(len(text)) * 12 + 8 * n + 64          # NFKC expansions x byte fallback
    pcap = int(len(text)) * 80 + 64 * n + 256
    ids = np.empty(cap, dtype=np.int32)
    begin = np.empty(cap, dtype=np.uint32)
    end = np.empty(cap, dtype=np.uint32)
    id_offs = np.zeros(n + 1, dtype=np.uint64)
    blob = np.empty(pcap, dtype=np.uint8)
    poffs = np.zeros(cap + 1, dtype=np.uint64)
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                   C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    tot = fn(h, text.ctypes.data if len(text) else None, offs.ctypes.data, n, ids.ctypes.data, begin.ctypes.data,
             end.ctypes.data, cap, id_offs.ctypes.data, blob.ctypes.data, pcap, poffs.ctypes.data)
    if tot < 0:
        raise RuntimeError("encode_pieces failed: %d" % tot)
    return (ids[:tot].copy(), begin[:tot].copy(), end[:tot].copy(), id_offs, blob[:int(poffs[tot])].tobytes(),
            poffs[:tot + 1].copy())


def normalize_batch(fn, h, text, offs):
    """-> (normalized uint8, norm_offsets uint64[n + 1], n2o uint32[total + n])"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    cap = int(len(text)) * 20 + 8 * n + 64
    out = np.empty(cap, dtype=np.uint8)
    no = np.zeros(n + 1, dtype=np.uint64)
    n2o = np.zeros(cap + n + 1, dtype=np.uint32)
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    tot = fn(h, text.ctypes.data if len(text) else None, offs.ctypes.data, n, out.ctypes.data, cap, no.ctypes.data,
             n2o.ctypes.data)
    if tot < 0:
        raise RuntimeError("normalize_batch failed: %d" % tot)
    return out[:tot].copy(), no, n2o[:tot + n].copy()


def compose_pieces(ids, nb, ne, id_offs, norm, norm_offs, id_to_piece, literal):
    """Piece strings the way the host facades build them: normalized[nb, ne) of the sentence, or the piece name where
    `literal(id)` (byte pieces, bos / eos, unknown under the unk_piece option).  -> (blob, piece_offsets)"""
    norm = bytes(norm)
    parts = []
    n = len(id_offs) - 1
    for s in range(n):
        base = int(norm_offs[s])
        for k in range(int(id_offs[s]), int(id_offs[s + 1])):
            t = int(ids[k])
            parts.append(id_to_piece(t) if literal(t) else norm[base + int(nb[k]):base + int(ne[k])])
    poffs = np.zeros(len(parts) + 1, dtype=np.uint64)
    if parts:
        poffs[1:] = np.cumsum([len(p) for p in parts])
    return b"".join(parts), poffs
