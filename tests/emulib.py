"""The product's C ABI (csrc/api.cc, unchanged) and device bodies run on the CPU: tests/emu/libspmx_emu.so is api.cc +
the kernels compiled against a lock-step model of the wavefront (tests/emu/wave_emu.h) and a host-memory stand-in for
the HIP runtime (tests/emu/fakehip).  TEST INFRASTRUCTURE ONLY: the product binds libspmx.so, never this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libspmx_emu.so")

# a class table with a tiny first class, so that short test inputs exercise the overflow / escalation paths too
SMALL_CLASSES = "24:40,192:448,576:1280,1536:3328,4096:8704,16384:32768,65536:98304"

_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", EMU_DIR])
        from sentencepiece_amd import _capi
        _lib = _capi.bind(EMU_SO)
    return _lib


class EmuLib:
    def __init__(self):
        self.lib = lib()

    def load(self, model_bytes, cus=2, classes=SMALL_CLASSES, env=None):
        """env: SPMX_* switches read at load (restored afterwards)."""
        e = dict(env or {})
        e.setdefault("SPMX_EMU_CUS", str(cus))
        e.setdefault("SPMX_FORCE_WORD_DP", "1")     # small test batches: the word form's second pass always runs
        e.setdefault("SPMX_UNI_WAVE_MAX", "0")      # ... and the staged classes keep the lane-per-sentence kernels (the
                                                    # wave-cooperative form has tests of its own; documents always take it)
        if classes:
            e.setdefault("SPMX_CLASSES", classes)
        old = {k: os.environ.get(k) for k in e}
        os.environ.update(e)
        try:
            return EmuHandle(self.lib, model_bytes)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def split_lines(self, data, grid=3):
        """bytes -> (packed text bytes, offsets uint64[n + 1]) through the device splitter."""
        from tests import fixtures
        h = self.load(fixtures.model_blob("test_model"))
        n = len(data)
        raw = np.zeros(n + 64, dtype=np.uint8)
        shift = (-raw.ctypes.data) & 15
        buf = raw[shift:shift + n + 16]
        buf[:n] = np.frombuffer(data, dtype=np.uint8)
        buf[n:] = 0x0A            # padding must not count
        text = np.full(n + 32, 0xCD, dtype=np.uint8)
        offs = np.full(data.count(b"\n") + 3, 0xCDCDCDCD, dtype=np.uint64)
        nl, tb = C.c_uint64(0), C.c_uint64(0)
        rc = self.lib.spmx_split_lines_device(h.sp._h, buf.ctypes.data, n, text.ctypes.data, len(text), offs.ctypes.data,
                                              len(offs), None, C.byref(nl), C.byref(tb))
        assert rc == 0, self.lib.spmx_last_error(None)
        assert (text[tb.value:] == 0xCD).all(), "write past the packed text"
        assert (offs[nl.value + 1:] == 0xCDCDCDCD).all(), "write past the offsets"
        return text[:tb.value].tobytes(), offs[:nl.value + 1].copy()


class GpuLib:
    """The same test-facing interface over the PRODUCT library (sentencepiece_amd/libspmx.so) on a real GPU: what lets a
    test written against EmuLib run as its own -m gpu twin (tests/test_processor_kats.py, tests/test_normalizer_kats.py)."""

    def __init__(self):
        from sentencepiece_amd import _capi
        self.lib = _capi.lib()

    def load(self, model_bytes, cus=None, classes=None, env=None):
        e = dict(env or {})
        old = {k: os.environ.get(k) for k in e}
        os.environ.update(e)
        try:
            return EmuHandle(self.lib, model_bytes)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


def backend(request_param):
    """'emu' -> EmuLib(), 'gpu' -> GpuLib(): for fixtures parametrized over both."""
    return GpuLib() if request_param == "gpu" else EmuLib()


BACKENDS = ["emu", "gpu"]


class EmuHandle:
    """The test-facing wrapper: a SentencePieceProcessor bound to the emulated library + the packed calls."""

    def __init__(self, lib_, model_bytes):
        from sentencepiece_amd.processor import SentencePieceProcessor
        self.lib = lib_
        self.sp = SentencePieceProcessor(model_proto=model_bytes, _lib=lib_)
        self.sp.SetProfiling(True)
        self.status = 0
        self.sent_status = None

    def flags(self):
        return int(self.lib.spmx_model_flags(self.sp._h))

    def path(self):
        """dict(hard, overflow, long, failed): how many sentences of the last encode took which way."""
        return self.sp.LastProfile()["path"]

    def set_encode_extra_options(self, opts):
        self.sp.SetEncodeExtraOptions(opts)

    def set_vocabulary(self, pieces):
        self.sp.SetVocabulary(pieces)

    def reset_vocabulary(self):
        self.sp.ResetVocabulary()

    def encode_batch(self, text, offs, grid=None):
        ids, io, st, failed = self.sp.EncodePackedEx(text, offs)
        self.sent_status = st
        self.status = int(failed)
        return ids, io

    def nbest(self, text, offs, nbest, grid=None):
        """NBestEncode of every sentence through the device kernels -> per sentence a list of (ids list, score)."""
        ids, io, sc, ro = self.sp.NBestPacked(np.ascontiguousarray(text, dtype=np.uint8),
                                              np.ascontiguousarray(offs, dtype=np.uint64), nbest)
        out = []
        for s in range(len(offs) - 1):
            out.append([(ids[int(io[k]):int(io[k + 1])].tolist(), float(sc[k])) for k in range(int(ro[s]), int(ro[s + 1]))])
        return out

    def normalize_batch(self, text, offs, grid=None):
        """-> (normalized uint8, norm_offsets, n2o) through the device normalize kernels."""
        return self.sp.NormalizePacked(text, offs, with_offsets=True)

    def encode_spans(self, text, offs, grid=None, norm_spans=False):
        """-> (ids, begin, end, id_offsets[, nbegin, nend]) through the device kernels (encode in spans form + align)."""
        return self.sp.EncodeSpansPacked(text, offs, norm_spans=norm_spans)

    def decode_batch(self, ids, id_offsets, grid=None):
        """Device decode kernels under the emulator -> (text uint8, text_offsets uint64); raises on a bad id."""
        return self.sp.DecodePacked(ids, id_offsets)
