"""ctypes binding of oracle/liboracle.so (the plain-C restatement).

TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")


def build():
    if not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ROOT, "oracle", "spm_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


class OracleLib:
    def __init__(self):
        build()
        self.lib = lib = C.CDLL(ORACLE_SO)
        lib.oracle_load.restype = C.c_void_p
        lib.oracle_load.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
        lib.oracle_free.argtypes = [C.c_void_p]
        lib.oracle_set_encode_extra_options.argtypes = [C.c_void_p, C.c_char_p]
        lib.oracle_set_vocabulary.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        lib.oracle_reset_vocabulary.argtypes = [C.c_void_p]
        lib.oracle_normalize.restype = C.c_int64
        lib.oracle_normalize.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        lib.oracle_encode.restype = C.c_int64
        lib.oracle_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        lib.oracle_encode_batch.restype = C.c_int64
        lib.oracle_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                            C.c_uint64, C.c_void_p]
        lib.oracle_piece_size.argtypes = [C.c_void_p]
        lib.oracle_decode_batch.restype = C.c_int64
        lib.oracle_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
        lib.oracle_model_type.argtypes = [C.c_void_p]

    def load(self, model_bytes):
        err = C.create_string_buffer(256)
        h = self.lib.oracle_load(model_bytes, len(model_bytes), err, 256)
        if not h:
            raise RuntimeError("oracle_load: " + err.value.decode())
        return OracleHandle(self.lib, h)


class OracleHandle:
    def __init__(self, lib, h):
        self.lib, self.h = lib, h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_free(self.h)
            self.h = None

    def set_encode_extra_options(self, opts):
        rc = self.lib.oracle_set_encode_extra_options(self.h, opts.encode())
        if rc:
            raise RuntimeError("bad extra options")

    def set_vocabulary(self, pieces):
        blob = "\n".join(pieces).encode()
        self.lib.oracle_set_vocabulary(self.h, blob, len(blob))

    def reset_vocabulary(self):
        self.lib.oracle_reset_vocabulary(self.h)

    def model_type(self):
        return self.lib.oracle_model_type(self.h)

    def encode(self, text):
        if isinstance(text, str):
            text = text.encode()
        cap = 4 * len(text) + 16
        out = np.empty(cap, dtype=np.int32)
        n = self.lib.oracle_encode(self.h, text, len(text), out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("oracle_encode failed: %d" % n)
        return out[:n].copy()

    def normalize(self, text):
        if isinstance(text, str):
            text = text.encode()
        cap = 32 * len(text) + 64
        out = np.empty(cap, dtype=np.uint8)
        n = self.lib.oracle_normalize(self.h, text, len(text), out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("oracle_normalize failed: %d" % n)
        return out[:n].tobytes()

    def encode_batch(self, text, offs):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        cap = int(len(text)) * 2 + 8 * n + 64
        ids = np.empty(cap, dtype=np.int32)
        id_offs = np.empty(n + 1, dtype=np.uint64)
        tp = text.ctypes.data if len(text) else None
        tot = self.lib.oracle_encode_batch(self.h, tp, offs.ctypes.data, n, ids.ctypes.data, cap,
                                           id_offs.ctypes.data)
        if tot < -1:
            cap = -tot - 2
            ids = np.empty(cap, dtype=np.int32)
            tot = self.lib.oracle_encode_batch(self.h, tp, offs.ctypes.data, n, ids.ctypes.data, cap,
                                               id_offs.ctypes.data)
        if tot < 0:
            raise RuntimeError("oracle_encode_batch failed")
        return ids[:tot].copy(), id_offs


def _oracle_decode_batch(self, ids, id_offsets):
    """Oracle Decode(ids) per sentence -> (text uint8, text_offsets uint64); raises on an invalid id."""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    id_offsets = np.ascontiguousarray(id_offsets, dtype=np.uint64)
    n = len(id_offsets) - 1
    cap = int(len(ids)) * 64 + 64
    text = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    tot = self.lib.oracle_decode_batch(self.h, ids.ctypes.data if len(ids) else None, id_offsets.ctypes.data, n,
                                       text.ctypes.data, cap, offs.ctypes.data)
    if tot < 0:
        raise RuntimeError("oracle_decode_batch failed: %d" % tot)
    return text[:tot].copy(), offs


OracleHandle.decode_batch = _oracle_decode_batch


def _oracle_encode_spans(self, text, offs):
    """Encode(input, SentencePieceText*) per sentence -> (ids, begin, end, id_offsets)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    cap = int(len(text)) * 3 + 8 * n + 64
    ids = np.empty(cap, dtype=np.int32)
    begin = np.empty(cap, dtype=np.uint32)
    end = np.empty(cap, dtype=np.uint32)
    id_offs = np.zeros(n + 1, dtype=np.uint64)
    fn = self.lib.oracle_encode_spans_batch
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    tot = fn(self.h, text.ctypes.data if len(text) else None, offs.ctypes.data, n, ids.ctypes.data, begin.ctypes.data,
             end.ctypes.data, cap, id_offs.ctypes.data)
    if tot < -1:           # -(needed) - 2: expansions + byte fallback can exceed the first guess
        cap = -tot - 2
        ids = np.empty(cap, dtype=np.int32)
        begin = np.empty(cap, dtype=np.uint32)
        end = np.empty(cap, dtype=np.uint32)
        tot = fn(self.h, text.ctypes.data if len(text) else None, offs.ctypes.data, n, ids.ctypes.data, begin.ctypes.data,
                 end.ctypes.data, cap, id_offs.ctypes.data)
    if tot < 0:
        raise RuntimeError("oracle_encode_spans_batch failed: %d" % tot)
    return ids[:tot].copy(), begin[:tot].copy(), end[:tot].copy(), id_offs


OracleHandle.encode_spans = _oracle_encode_spans


def _oracle_encode_pieces(self, text, offs):
    from tests import pieceslib
    return pieceslib.encode_pieces(self.lib.oracle_encode_pieces_batch, self.h, text, offs)


def _oracle_normalize_batch(self, text, offs):
    from tests import pieceslib
    return pieceslib.normalize_batch(self.lib.oracle_normalize_batch, self.h, text, offs)


OracleHandle.encode_pieces = _oracle_encode_pieces
OracleHandle.normalize_batch = _oracle_normalize_batch
