"""include/spmx_reference_binding.h -- the subclass INTEGRATION.md section 2 shows -- compiled against the reference's
own header and objects (oracle/_ref/obj, made by oracle/Makefile from the sources where they lie) and the product's C
ABI, then driven through a ``sentencepiece::SentencePieceProcessor*`` with spm_encode's loop next to the unmodified
base class (tests/cpp/ref_binding_test.cc).  The CPU run links the emulated library; the GPU run uses the binary that
``__graft_entry__.build()`` / this module made in the build container against libspmx.so (the GPU box has no
/root/reference to compile against)."""
import glob
import os
import subprocess

import pytest

from tests import fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BIN = os.path.join(ROOT, "tests", "cpp", "ref_binding_test")


def build(emu):
    """-> path of the binary, or None where the reference tree / its objects are not there."""
    objs = sorted(glob.glob(os.path.join(ROOT, "oracle", "_ref", "obj", "**", "*.o"), recursive=True))
    if not os.path.isdir(os.path.join(REF, "src")) or not objs:
        return None
    out = BIN + ("_emu" if emu else "")
    lib = os.path.join(ROOT, "tests", "emu") if emu else os.path.join(ROOT, "sentencepiece_amd")
    if emu:
        from tests import emulib
        emulib.lib()
    so = os.path.join(lib, "libspmx_emu.so" if emu else "libspmx.so")
    if not os.path.exists(so):
        return None
    srcs = [os.path.join(ROOT, "tests", "cpp", "ref_binding_test.cc"), os.path.join(ROOT, "include", "spmx_reference_binding.h"),
            os.path.join(ROOT, "include", "spmx.h"), so]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(p) for p in srcs):
        inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle", "_ref"), "-I" + REF, "-I" + REF + "/src",
               "-I" + REF + "/src/builtin_pb", "-I" + REF + "/third_party", "-I" + REF + "/third_party/protobuf-lite"]
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-pthread", "-DHAVE_PTHREAD=1", "-D_USE_INTERNAL_STRING_VIEW"] + inc +
                              ["-o", out, srcs[0]] + objs + ["-L" + lib, "-lspmx_emu" if emu else "-lspmx", "-Wl,-rpath," + lib, "-lpthread"])
    return out


CASES = [("test_model", ""), ("test_model", "bos:eos"), ("bpe1k", ""), ("uni1k_bf", "reverse")]


@pytest.mark.parametrize("model,opts", CASES)
def test_reference_binding_emulated(model, opts, tmp_path):
    b = build(emu=True)
    if b is None:
        pytest.skip("no reference tree / compiled reference objects here")
    text = os.path.join(str(tmp_path), "lines.txt")
    with open(os.path.join(fixtures.GOLDEN, "botchan.txt"), "rb") as f:
        lines = f.read().split(b"\n")[:400]
    with open(text, "wb") as f:
        f.write(b"\n".join(lines) + b"\n")
    args = [b, os.path.join(fixtures.GOLDEN, model + ".model"), text] + ([opts] if opts else [])
    out = subprocess.run(args, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("OK 400 ")


@pytest.mark.gpu
@pytest.mark.parametrize("model,opts", CASES)
def test_reference_binding_gpu(model, opts):
    b = build(emu=False) or (BIN if os.path.exists(BIN) else None)
    if b is None:
        pytest.skip("tests/cpp/ref_binding_test was not prebuilt (no reference tree in the build container)")
    args = [b, os.path.join(fixtures.GOLDEN, model + ".model"), os.path.join(fixtures.GOLDEN, "botchan.txt")] + ([opts] if opts else [])
    out = subprocess.run(args, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("OK 4288 ")
