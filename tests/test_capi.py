"""The C-ABI library loads and exports every symbol include/spmx.h declares;
without a GPU it refuses to create a handle (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from tests import fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "spmx.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spmx_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported():
    from sentencepiece_amd import _capi
    lib = _capi.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(s[0] for s in _capi.SYMBOLS) == names


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sentencepiece_amd import _capi
    lib = _capi.lib()
    h = C.c_void_p()
    blob = fixtures.model_blob("test_model")
    rc = lib.spmx_create(blob, len(blob), 0, C.byref(h))
    assert rc == 14 and not h.value          # util::StatusCode::kUnavailable
    assert b"no HIP device" in lib.spmx_last_error(None)
    from sentencepiece_amd.processor import SentencePieceProcessor
    with pytest.raises(RuntimeError):
        SentencePieceProcessor(model_proto=blob)


def test_product_does_not_touch_oracle():
    """Nothing under sentencepiece_amd/ may import, link or call oracle/ or the emulator."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "sentencepiece_amd")):
        for fn in files:
            if not fn.endswith((".py", ".cc", ".h", ".hip", "Makefile")):
                continue
            with open(os.path.join(d, fn), errors="replace") as f:
                s = f.read()
            if re.search(r"oracle|wave_emu|emu_lib|refshim|libspm_ref", s):
                bad.append(fn)
    assert bad == []


def _handle_info_checks(load):
    """spmx_handle_info: table bytes and load time per handle; the sentence-per-lane kernels' memo tiers are built only
    for a handle that asks for those kernels (tables.cc legacy_tiers)."""
    from tests import fixtures
    blob = fixtures.model_blob("uni32k")
    a = load(blob, {"SPMX_FORCE_WORD_DP": "0"}).sp.HandleInfo()      # (the emulator's loader sets it to 1 by default: the DP pass reads the tiers)
    b = load(blob, {"SPMX_FORCE_WORD_DP": "0", "SPMX_WORD_WAVE": "0"}).sp.HandleInfo()
    assert a["load_ms"] > 0.0 and b["load_ms"] > 0.0
    assert 1 << 20 < a["table_bytes"] < 1 << 28, a
    assert b["table_bytes"] > a["table_bytes"] + (1 << 20), (a, b)      # umemo16 + umemo: megabytes for a 32k vocabulary


def test_emu_handle_info():
    from tests import emulib
    lib = emulib.EmuLib()
    _handle_info_checks(lambda blob, env: lib.load(blob, env=env))


@pytest.mark.gpu
def test_gpu_handle_info():
    from tests import emulib
    lib = emulib.GpuLib()
    _handle_info_checks(lambda blob, env: lib.load(blob, env=env))
