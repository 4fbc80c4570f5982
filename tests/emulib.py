"""ctypes binding of tests/emu/libspmx_emu.so: the product's device code run on
the CPU under a lock-step wavefront model.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libspmx_emu.so")


class EmuLib:
    def __init__(self):
        subprocess.check_call(["make", "-s", "-C", EMU_DIR])
        self.lib = lib = C.CDLL(EMU_SO)
        lib.emu_load.restype = C.c_void_p
        lib.emu_load.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
        lib.emu_free.argtypes = [C.c_void_p]
        lib.emu_set_encode_extra_options.argtypes = [C.c_void_p, C.c_char_p]
        lib.emu_set_vocabulary.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        lib.emu_reset_vocabulary.argtypes = [C.c_void_p]
        lib.emu_encode_batch.restype = C.c_int64
        lib.emu_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                         C.c_void_p, C.c_int, C.c_void_p]
        lib.emu_collectives.restype = C.c_uint64
        lib.emu_decode_batch.restype = C.c_int64
        lib.emu_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                         C.c_void_p, C.c_int, C.c_void_p]
        lib.emu_encode_spans_batch.restype = C.c_int64
        lib.emu_encode_spans_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
        lib.emu_normalize_batch.restype = C.c_int64
        lib.emu_normalize_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.emu_split_lines.restype = C.c_int64
        lib.emu_split_lines.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.emu_flags.restype = C.c_uint32
        lib.emu_flags.argtypes = [C.c_void_p]
        lib.emu_fast_kept.restype = C.c_uint64
        lib.emu_fast_handed.restype = C.c_uint64
        lib.emu_wave_handed.restype = C.c_uint64

    def split_lines(self, data, grid=3):
        """bytes -> (packed text bytes, offsets uint64[n + 1]) through the device splitter."""
        n = len(data)
        raw = np.zeros(n + 64, dtype=np.uint8)
        shift = (-raw.ctypes.data) & 15
        buf = raw[shift:shift + n + 16]
        buf[:n] = np.frombuffer(data, dtype=np.uint8)
        buf[n:] = 0x0A            # padding must not count
        text = np.full(n + 32, 0xCD, dtype=np.uint8)
        offs = np.full(data.count(b"\n") + 3, 0xCDCDCDCD, dtype=np.uint64)
        tb = C.c_uint64(0)
        lines = self.lib.emu_split_lines(buf.ctypes.data, n, text.ctypes.data, offs.ctypes.data, grid, C.byref(tb))
        assert (text[tb.value:] == 0xCD).all(), "write past the packed text"
        assert (offs[lines + 1:] == 0xCDCDCDCD).all(), "write past the offsets"
        return text[:tb.value].tobytes(), offs[:lines + 1].copy()

    def load(self, model_bytes):
        err = C.create_string_buffer(512)
        h = self.lib.emu_load(model_bytes, len(model_bytes), err, 512)
        if not h:
            raise RuntimeError("emu_load: " + err.value.decode())
        return EmuHandle(self.lib, h)


class EmuHandle:
    def __init__(self, lib, h):
        self.lib, self.h = lib, h
        self.status = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.emu_free(self.h)
            self.h = None

    def flags(self):
        return int(self.lib.emu_flags(self.h))

    def fast_split(self):
        """(sentences the FAST tile kernel kept, sentences it handed to the GENERAL kernel) in the last call."""
        return int(self.lib.emu_fast_kept()), int(self.lib.emu_fast_handed())

    def wave_handed(self):
        """BPE: sentences the lane form handed to the sentence-per-wave kernel in the last call."""
        return int(self.lib.emu_wave_handed())

    def set_encode_extra_options(self, opts):
        rc = self.lib.emu_set_encode_extra_options(self.h, opts.encode())
        if rc:
            raise RuntimeError("bad extra options (%d)" % rc)

    def set_vocabulary(self, pieces):
        blob = "\n".join(pieces).encode()
        self.lib.emu_set_vocabulary(self.h, blob, len(blob))

    def reset_vocabulary(self):
        self.lib.emu_reset_vocabulary(self.h)

    def encode_batch(self, text, offs, grid=3):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        cap = int(len(text)) * 2 + 8 * n + 64
        ids = np.empty(cap, dtype=np.int32)
        id_offs = np.zeros(n + 1, dtype=np.uint64)
        st = C.c_uint32(0)
        tp = text.ctypes.data if len(text) else None
        tot = self.lib.emu_encode_batch(self.h, tp, offs.ctypes.data, n, ids.ctypes.data, cap, id_offs.ctypes.data,
                                        grid, C.byref(st))
        self.status = st.value
        if tot < 0:
            raise RuntimeError("emu_encode_batch failed: %d status %d" % (tot, st.value))
        return ids[:tot].copy(), id_offs


def _emu_nbest(self, text, offs, nbest, grid=2):
    """NBestEncode of every sentence through the device kernels -> per sentence a list of (ids list, score)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    cap = (int(len(text)) * 4 + 16 * n + 64) * nbest
    ids = np.empty(cap, dtype=np.int32)
    id_offs = np.zeros(n * nbest + 2, dtype=np.uint64)
    scores = np.zeros(n * nbest + 1, dtype=np.float32)
    res_offs = np.zeros(n + 1, dtype=np.uint64)
    st = C.c_uint32(0)
    fn = self.lib.emu_nbest_batch
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    r = fn(self.h, text.ctypes.data if len(text) else None, offs.ctypes.data, n, nbest, ids.ctypes.data, cap,
           id_offs.ctypes.data, scores.ctypes.data, res_offs.ctypes.data, grid, C.byref(st))
    self.status = st.value
    if r < 0:
        raise RuntimeError("emu_nbest_batch failed: %d status %d" % (r, st.value))
    out = []
    for s in range(n):
        out.append([(ids[int(id_offs[k]):int(id_offs[k + 1])].tolist(), float(scores[k]))
                    for k in range(int(res_offs[s]), int(res_offs[s + 1]))])
    return out


EmuHandle.nbest = _emu_nbest


def _emu_normalize_batch(self, text, offs, grid=3):
    """-> (normalized uint8, norm_offsets, n2o) through the device normalize kernels."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    cap = int(len(text)) * 20 + 8 * n + 64
    out = np.full(cap, 0xCD, dtype=np.uint8)
    no = np.zeros(n + 1, dtype=np.uint64)
    n2o = np.full(cap + n + 1, 0xCDCDCDCD, dtype=np.uint32)
    st = C.c_uint32(0)
    tot = self.lib.emu_normalize_batch(self.h, text.ctypes.data if len(text) else None, offs.ctypes.data, n,
                                       out.ctypes.data, cap, no.ctypes.data, n2o.ctypes.data, grid, C.byref(st))
    self.status = st.value
    if tot < 0:
        raise RuntimeError("emu_normalize_batch failed: %d status %d" % (tot, st.value))
    assert (out[tot:] == 0xCD).all() and (n2o[tot + n:] == 0xCDCDCDCD).all(), "write past the end"
    return out[:tot].copy(), no, n2o[:tot + n].copy()


EmuHandle.normalize_batch = _emu_normalize_batch


def _emu_encode_spans(self, text, offs, grid=3, norm_spans=False):
    """-> (ids, begin, end, id_offsets[, nbegin, nend]) through the device kernels (encode in spans form + align)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    cap = int(len(text)) * 12 + 8 * n + 64          # NFKC expansions x byte fallback
    ids = np.empty(cap, dtype=np.int32)
    begin = np.full(cap, 0xCDCDCDCD, dtype=np.uint32)
    end = np.full(cap, 0xCDCDCDCD, dtype=np.uint32)
    id_offs = np.zeros(n + 1, dtype=np.uint64)
    st = C.c_uint32(0)
    nb = np.full(cap, 0xCDCDCDCD, dtype=np.uint32)
    ne = np.full(cap, 0xCDCDCDCD, dtype=np.uint32)
    tot = self.lib.emu_encode_spans_batch(self.h, text.ctypes.data if len(text) else None, offs.ctypes.data, n,
                                          ids.ctypes.data, begin.ctypes.data, end.ctypes.data, cap, id_offs.ctypes.data,
                                          grid, C.byref(st), nb.ctypes.data if norm_spans else None,
                                          ne.ctypes.data if norm_spans else None)
    self.status = st.value
    if tot < 0:
        raise RuntimeError("emu_encode_spans_batch failed: %d status %d" % (tot, st.value))
    if norm_spans:
        return ids[:tot].copy(), begin[:tot].copy(), end[:tot].copy(), id_offs, nb[:tot].copy(), ne[:tot].copy()
    return ids[:tot].copy(), begin[:tot].copy(), end[:tot].copy(), id_offs


EmuHandle.encode_spans = _emu_encode_spans


def _emu_decode_batch(self, ids, id_offsets, grid=3):
    """Device decode kernels under the emulator -> (text uint8, text_offsets uint64); raises on a bad id."""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    id_offsets = np.ascontiguousarray(id_offsets, dtype=np.uint64)
    n = len(id_offsets) - 1
    cap = int(len(ids)) * 64 + 64
    text = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    st = C.c_uint32(0)
    tot = self.lib.emu_decode_batch(self.h, ids.ctypes.data if len(ids) else None, id_offsets.ctypes.data, n,
                                    text.ctypes.data, cap, offs.ctypes.data, grid, C.byref(st))
    self.status = st.value
    if tot < 0:
        raise RuntimeError("emu_decode_batch failed: %d status %d" % (tot, st.value))
    return text[:tot].copy(), offs


EmuHandle.decode_batch = _emu_decode_batch
