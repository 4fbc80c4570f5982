"""The host-buffer side of the boundary: the chunk pipeline of spmx_encode_batch, the
(pointer, length) form behind EncodeBatch(vector<string_view>), concurrent callers on one handle, and the corpus file
tool -- on the CPU with the device emulated (tests/emulib.py), on the GPU with -m gpu."""
import ctypes as C
import hashlib
import os
import threading

import numpy as np
import pytest

from tests import fixtures

# the survey's known answer: `spm_encode --model=test_model.model --output_format=id < botchan.txt | md5sum`
BOTCHAN_ID_MD5 = "ff197d02d69c7695fccfec3bac27bf7a"


@pytest.fixture(scope="module")
def emu():
    from tests import emulib
    return emulib.EmuLib()


def views_call(lib, handle, bufs):
    class V(C.Structure):
        _fields_ = [("data", C.c_char_p), ("len", C.c_uint64)]
    arr = (V * len(bufs))(*[V(b, len(b)) for b in bufs])
    p_ids, p_off, p_st = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nf = C.c_uint64(0)
    rc = lib.spmx_encode_batch_views(handle, arr, len(bufs), C.byref(p_ids), C.byref(p_off), C.byref(p_st), C.byref(nf))
    assert rc == 0, lib.spmx_last_error(None)
    n = len(bufs)
    io = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    ids = (np.ctypeslib.as_array(C.cast(p_ids, C.POINTER(C.c_int32)), shape=(int(io[n]),)).copy() if io[n] else np.zeros(0, np.int32))
    for p in (p_ids, p_off, p_st):
        lib.spmx_free(p)
    return ids, io


@pytest.mark.parametrize("model", ["uni32k", "bpe32k", "uni1k_bf"])
def test_emu_chunk_pipeline_and_views(model, emu, oracle, corpora):
    """A batch big enough for the chunk pipeline (chunks of 1024 sentences here, 3 workers): the same CSR as the oracle's,
    through the packed form and through the (pointer, length) form."""
    blob = fixtures.model_blob(model)
    h = emu.load(blob, classes=None, env={"SPMX_HOST_CHUNK": "1024", "SPMX_HOST_THREADS": "3"})
    o = oracle.load(blob)
    text, offs = fixtures.head(*corpora["synth20k"], 6500)
    ids, io = h.encode_batch(text, offs)
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    assert not h.sent_status.any()
    tb = np.asarray(text).tobytes()
    bufs = [tb[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    vids, vio = views_call(h.lib, h.sp._h, bufs)
    np.testing.assert_array_equal(vio, oio)
    np.testing.assert_array_equal(vids, oids)


def test_emu_concurrent_callers(emu, oracle, corpora):
    """Encode is const in the reference and may be called from several threads at once: here every call leases its own
    workspace and stream."""
    blob = fixtures.model_blob("test_model")
    h, o = emu.load(blob), oracle.load(blob)
    parts = [fixtures.head(*corpora[name], k) for name, k in (("botchan", 400), ("edge", 10 ** 6), ("synth20k", 500), ("mixed2k", 60))]
    want = [o.encode_batch(t, of) for t, of in parts]
    got = [None] * len(parts)

    def run(i):
        for _ in range(3):
            got[i] = h.sp.EncodePacked(*parts[i])
    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(parts))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for (ids, io), (oids, oio) in zip(got, want):
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


def test_emu_encode_file(emu, tmp_path):
    """spmx_encode_file, format "id": the bytes `spm_encode --output_format=id` writes for botchan.txt with the bundled
    model (95,515 tokens, md5 ff197d02...); format "bin": the same ids, flat."""
    h = emu.load(fixtures.model_blob("test_model"), classes=None)
    src = os.path.join(fixtures.GOLDEN, "botchan.txt")
    out = str(tmp_path / "ids.txt")
    ns, ni = h.sp.EncodeFile(src, out, "id")
    assert (ns, ni) == (4288, 95515)
    data = open(out, "rb").read()
    assert hashlib.md5(data).hexdigest() == BOTCHAN_ID_MD5
    outb = str(tmp_path / "ids.bin")
    assert h.sp.EncodeFile(src, outb, "bin") == (4288, 95515)
    ids = np.fromfile(outb, dtype=np.int32)
    io = np.fromfile(outb + ".idx", dtype=np.uint64)
    assert len(io) == 4289 and int(io[-1]) == 95515 == len(ids)
    lines = data.decode().split("\n")[:-1]
    assert [int(x) for x in lines[7].split()] == ids[int(io[7]):int(io[8])].tolist()


@pytest.mark.gpu
def test_gpu_encode_file_and_cli(tmp_path):
    """The same known answer on the GPU, through the Python wrapper and through the spmx_encode command line."""
    import subprocess
    from sentencepiece_amd.processor import SentencePieceProcessor
    model = os.path.join(fixtures.GOLDEN, "test_model.model")
    src = os.path.join(fixtures.GOLDEN, "botchan.txt")
    sp = SentencePieceProcessor(model_file=model)
    out = str(tmp_path / "ids.txt")
    assert sp.EncodeFile(src, out, "id") == (4288, 95515)
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == BOTCHAN_ID_MD5
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sentencepiece_amd", "spmx_encode")
    r = subprocess.run([exe, "--model=" + model, "--output_format=id", src], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert hashlib.md5(r.stdout).hexdigest() == BOTCHAN_ID_MD5


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["uni32k", "bpe32k"])
def test_gpu_chunk_pipeline(model, oracle):
    """2.2 M sentences through the host-buffer form (chunk pipeline: 8 workers) and the (pointer, length) form: a strided
    sample equals the oracle's ids, and both forms agree everywhere."""
    from sentencepiece_amd import synth
    from sentencepiece_amd.processor import SentencePieceProcessor
    blob = fixtures.model_blob(model)
    sp = SentencePieceProcessor(model_proto=blob)
    text, offs = synth.ascii_corpus(2_200_000, seed=5)
    ids, io, st, failed = sp.EncodePackedEx(text, offs)
    assert failed == 0 and not st.any()
    pick = np.arange(0, len(offs) - 1, 997)
    pt, po = synth.gather_packed(text, offs, pick)
    oids, oio = oracle.load(blob).encode_batch(pt, po)
    cnt = np.diff(io.astype(np.int64))
    np.testing.assert_array_equal(cnt[pick], np.diff(oio.astype(np.int64)))
    got = np.concatenate([ids[int(io[i]):int(io[i + 1])] for i in pick])
    np.testing.assert_array_equal(got, oids)
    d_ids, d_io, total = None, None, None
    import torch
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    d_ids, d_io, total = sp.EncodeDevice(d_text, d_offs)
    np.testing.assert_array_equal(d_io.cpu().numpy().astype(np.uint64), io)
    np.testing.assert_array_equal(d_ids[:total].cpu().numpy(), ids)


def test_emu_encode_batch_multi(emu, oracle, corpora):
    """spmx_encode_batch_multi: three handles (three GPUs of a node; here three emulated devices) share the chunks of one
    batch and fill one CSR."""
    blob = fixtures.model_blob("uni32k")
    env = {"SPMX_HOST_CHUNK": "1024", "SPMX_HOST_THREADS": "2"}
    hs = [emu.load(blob, classes=None, env=env) for _ in range(3)]
    text, offs = fixtures.head(*corpora["synth20k"], 9000)
    text, offs = np.ascontiguousarray(text), np.ascontiguousarray(offs, dtype=np.uint64)
    arr = (C.c_void_p * 3)(*[h.sp._h for h in hs])
    p_ids, p_off, p_st = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nf = C.c_uint64(0)
    n = len(offs) - 1
    lib = hs[0].lib
    rc = lib.spmx_encode_batch_multi(arr, 3, text.ctypes.data, offs.ctypes.data, n, C.byref(p_ids), C.byref(p_off), C.byref(p_st), C.byref(nf))
    assert rc == 0, lib.spmx_last_error(None)
    io = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    ids = np.ctypeslib.as_array(C.cast(p_ids, C.POINTER(C.c_int32)), shape=(int(io[n]),)).copy()
    for p in (p_ids, p_off, p_st):
        lib.spmx_free(p)
    oids, oio = oracle.load(blob).encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)


def test_emu_degenerate_batches(emu, oracle):
    """Empty batches, empty sentences and null pointers through every host form: the reference's batch is a loop of
    Encode calls, so n = 0 gives an empty CSR and an empty sentence an empty id range (bos / eos when asked for)."""
    blob = fixtures.model_blob("test_model")
    h, o = emu.load(blob), oracle.load(blob)
    lib = h.lib
    # n = 0 through the packed form
    p_ids, p_off = C.c_void_p(), C.c_void_p()
    offs0 = np.zeros(1, dtype=np.uint64)
    assert lib.spmx_encode_batch(h.sp._h, None, offs0.ctypes.data, 0, C.byref(p_ids), C.byref(p_off)) == 0
    assert np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_uint64)), shape=(1,))[0] == 0
    lib.spmx_free(p_ids); lib.spmx_free(p_off)
    # n = 0 and empty views
    ids, io = views_call(lib, h.sp._h, [])
    assert len(ids) == 0 and io.tolist() == [0]
    ids, io = views_call(lib, h.sp._h, [b"", b"a", b"", b""])
    want, wio = o.encode_batch(*__import__("sentencepiece_amd.synth", fromlist=["pack"]).pack([b"", b"a", b"", b""]))
    np.testing.assert_array_equal(io, wio)
    np.testing.assert_array_equal(ids, want)
    # empty sentences keep their bos / eos
    h.set_encode_extra_options("bos:eos")
    o.set_encode_extra_options("bos:eos")
    ids, io = views_call(lib, h.sp._h, [b"", b" ", b"x"])
    want, wio = o.encode_batch(*__import__("sentencepiece_amd.synth", fromlist=["pack"]).pack([b"", b" ", b"x"]))
    np.testing.assert_array_equal(io, wio)
    np.testing.assert_array_equal(ids, want)
    # null output containers are an error, not a crash
    assert lib.spmx_encode_batch(h.sp._h, None, offs0.ctypes.data, 0, None, None) != 0
    # the lattice entry points on an empty batch
    h.set_encode_extra_options("")
    for fn in (lambda: h.sp.SampleEncodePacked(np.zeros(0, np.uint8), offs0, -1, 0.1), lambda: h.sp.EncodeOriginalPacked(np.zeros(0, np.uint8), offs0)):
        ids, io = fn()
        assert len(ids) == 0 and io.tolist() == [0]
