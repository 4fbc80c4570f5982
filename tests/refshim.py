"""ctypes binding of oracle/_ref/libspm_ref.so (the compiled upstream reference).

TEST INFRASTRUCTURE ONLY.  Used by tests/, scripts/make_fixtures.py and
bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libspm_ref.so")


def available():
    return os.path.exists(REF_SO)


class RefLib:
    def __init__(self, path=REF_SO):
        self.lib = lib = C.CDLL(path)
        lib.spmref_load.restype = C.c_void_p
        lib.spmref_load.argtypes = [C.c_char_p, C.c_uint64]
        lib.spmref_free.argtypes = [C.c_void_p]
        lib.spmref_last_error.restype = C.c_char_p
        lib.spmref_last_error.argtypes = [C.c_void_p]
        lib.spmref_set_encode_extra_options.argtypes = [C.c_void_p, C.c_char_p]
        lib.spmref_set_decode_extra_options.argtypes = [C.c_void_p, C.c_char_p]
        lib.spmref_set_vocabulary.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        lib.spmref_reset_vocabulary.argtypes = [C.c_void_p]
        lib.spmref_encode.restype = C.c_int64
        lib.spmref_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        lib.spmref_normalize.restype = C.c_int64
        lib.spmref_normalize.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        lib.spmref_encode_batch.restype = C.c_int64
        lib.spmref_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                            C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        lib.spmref_encode_count.restype = C.c_int64
        lib.spmref_encode_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        lib.spmref_piece_size.argtypes = [C.c_void_p]
        lib.spmref_decode_batch.restype = C.c_int64
        lib.spmref_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]

    def load(self, model_bytes):
        h = self.lib.spmref_load(model_bytes, len(model_bytes))
        if not h:
            raise RuntimeError("reference failed to load the model")
        return RefHandle(self.lib, h)


class RefHandle:
    def __init__(self, lib, h):
        self.lib, self.h = lib, h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.spmref_free(self.h)
            self.h = None

    def set_encode_extra_options(self, opts):
        rc = self.lib.spmref_set_encode_extra_options(self.h, opts.encode())
        if rc:
            raise RuntimeError(self.lib.spmref_last_error(self.h).decode())

    def set_decode_extra_options(self, opts):
        rc = self.lib.spmref_set_decode_extra_options(self.h, opts.encode())
        if rc:
            raise RuntimeError(self.lib.spmref_last_error(self.h).decode())

    def set_vocabulary(self, pieces):
        blob = "\n".join(pieces).encode()
        rc = self.lib.spmref_set_vocabulary(self.h, blob, len(blob))
        if rc:
            raise RuntimeError(self.lib.spmref_last_error(self.h).decode())

    def reset_vocabulary(self):
        self.lib.spmref_reset_vocabulary(self.h)

    def piece_size(self):
        return self.lib.spmref_piece_size(self.h)

    def encode(self, text):
        if isinstance(text, str):
            text = text.encode()
        cap = 64 * len(text) + 64
        out = np.empty(cap, dtype=np.int32)
        n = self.lib.spmref_encode(self.h, text, len(text), out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("reference Encode failed: %d" % n)
        return out[:n].copy()

    def set_encoder_original(self):
        """unigram::Model::SetEncoderVersion(kOriginal) (src/unigram_model.h:176-186)."""
        self.lib.spmref_set_encoder_original.argtypes = [C.c_void_p]
        if self.lib.spmref_set_encoder_original(self.h) != 0:
            raise RuntimeError("not a unigram model")

    def sample_encode(self, text, nbest_size, alpha):
        fn = self.lib.spmref_sample_encode
        fn.restype = C.c_int64
        fn.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int, C.c_float, C.c_void_p, C.c_uint64]
        cap = 64 * len(text) + 64
        out = np.empty(cap, dtype=np.int32)
        n = fn(self.h, text, len(text), nbest_size, alpha, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("reference SampleEncode failed")
        return out[:n].tolist()

    def normalize(self, text):
        if isinstance(text, str):
            text = text.encode()
        cap = 32 * len(text) + 64
        out = np.empty(cap, dtype=np.uint8)
        n = self.lib.spmref_normalize(self.h, text, len(text), out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("reference Normalize failed: %d" % n)
        return out[:n].tobytes()

    def encode_batch(self, text, offs, threads=1):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        cap = int(len(text)) * 2 + 8 * n + 64
        ids = np.empty(cap, dtype=np.int32)
        id_offs = np.empty(n + 1, dtype=np.uint64)
        tp = text.ctypes.data if len(text) else None
        tot = self.lib.spmref_encode_batch(self.h, tp, offs.ctypes.data, n, ids.ctypes.data, cap,
                                           id_offs.ctypes.data, threads)
        if tot < -1:
            cap = -tot - 2
            ids = np.empty(cap, dtype=np.int32)
            tot = self.lib.spmref_encode_batch(self.h, tp, offs.ctypes.data, n, ids.ctypes.data, cap,
                                               id_offs.ctypes.data, threads)
        if tot < 0:
            raise RuntimeError("reference EncodeBatch failed")
        return ids[:tot].copy(), id_offs

    def encode_count(self, text, offs, threads=1):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        return self.lib.spmref_encode_count(self.h, text.ctypes.data, offs.ctypes.data,
                                            len(offs) - 1, threads)


def _decode_batch(fn, h, ids, id_offsets):
    import numpy as np
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    id_offsets = np.ascontiguousarray(id_offsets, dtype=np.uint64)
    n = len(id_offsets) - 1
    cap = int(len(ids)) * 64 + 64
    text = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    tot = fn(h, ids.ctypes.data if len(ids) else None, id_offsets.ctypes.data, n, text.ctypes.data, cap, offs.ctypes.data)
    if tot < 0:
        raise RuntimeError("decode_batch failed: %d" % tot)
    return text[:tot].copy(), offs


def _ref_decode_batch(self, ids, id_offsets):
    """Reference Decode(ids) per sentence -> (text uint8, text_offsets uint64)."""
    return _decode_batch(self.lib.spmref_decode_batch, self.h, ids, id_offsets)


RefHandle.decode_batch = _ref_decode_batch


def _ref_encode_spans(self, text, offs):
    """Encode(input, SentencePieceText*) per sentence -> (ids, begin, end, id_offsets)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    cap = int(len(text)) * 3 + 8 * n + 64
    ids = np.empty(cap, dtype=np.int32)
    begin = np.empty(cap, dtype=np.uint32)
    end = np.empty(cap, dtype=np.uint32)
    id_offs = np.zeros(n + 1, dtype=np.uint64)
    fn = self.lib.spmref_encode_spans_batch
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
        raise RuntimeError("spmref_encode_spans_batch failed: %d" % tot)
    return ids[:tot].copy(), begin[:tot].copy(), end[:tot].copy(), id_offs


RefHandle.encode_spans = _ref_encode_spans


def _spmref_encode_pieces(self, text, offs):
    from tests import pieceslib
    return pieceslib.encode_pieces(self.lib.spmref_encode_pieces_batch, self.h, text, offs)


def _spmref_normalize_batch(self, text, offs):
    from tests import pieceslib
    return pieceslib.normalize_batch(self.lib.spmref_normalize_batch, self.h, text, offs)


RefHandle.encode_pieces = _spmref_encode_pieces
RefHandle.normalize_batch = _spmref_normalize_batch


def _ref_decode_pieces(self, pieces):
    """Reference Decode(pieces) of one sentence (list of str / bytes) -> bytes."""
    pb = [p if isinstance(p, bytes) else p.encode("utf-8", "surrogateescape") for p in pieces]
    offs = np.zeros(len(pb) + 1, dtype=np.uint64)
    if pb:
        np.cumsum([len(x) for x in pb], out=offs[1:])
    blob = b"".join(pb)
    cap = len(blob) * 4 + 64 * (len(pb) + 1)
    out = C.create_string_buffer(cap)
    fn = self.lib.spmref_decode_pieces
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
    r = fn(self.h, blob, offs.ctypes.data, len(pb), out, cap)
    if r < 0:
        raise RuntimeError("spmref_decode_pieces failed: %d" % r)
    return out.raw[:r]


def _ref_get_score(self, id):
    fn = self.lib.spmref_get_score
    fn.restype = C.c_float
    fn.argtypes = [C.c_void_p, C.c_int]
    return float(fn(self.h, id))


RefHandle.decode_pieces = _ref_decode_pieces
RefHandle.get_score = _ref_get_score
