"""include/spmx_processor.h (the C++ SentencePieceProcessor-shaped facade over
the C ABI), compiled with g++ and linked against libspmx.so."""
import os
import subprocess

import numpy as np
import pytest

from tests import fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "facade_test")


def _build(emu=False):
    """emu: link the same caller against tests/emu/libspmx_emu.so (the product's api.cc over the CPU model of the
    wavefront) instead of libspmx.so: the facade's own code runs in the CPU suite too."""
    src = os.path.join(ROOT, "tests", "cpp", "facade_test.cc")
    lib = os.path.join(ROOT, "tests", "emu") if emu else os.path.join(ROOT, "sentencepiece_amd")
    out = BIN + ("_emu" if emu else "")
    if emu:
        from tests import emulib
        emulib.lib()                      # builds libspmx_emu.so
    so = os.path.join(lib, "libspmx_emu.so" if emu else "libspmx.so")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "spmx_processor.h")),
                 os.path.getmtime(os.path.join(ROOT, "include", "spmx.h")), os.path.getmtime(so))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", out, src, "-L" + lib,
                               "-lspmx_emu" if emu else "-lspmx", "-Wl,-rpath," + lib])
    return out


def test_facade_builds_and_reports_unavailable():
    import torch
    b = _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([b, os.path.join(fixtures.GOLDEN, "test_model.model"), "--expect-unavailable"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "Unavailable" in out.stdout


@pytest.mark.parametrize("model,opts,key", [("test_model", "bos:eos", "test_model__botchan__bos-eos"), ("bpe1k", "", "bpe1k__botchan")])
def test_facade_matches_golden_emulated(model, opts, key, golden_arrays, tmp_path):
    """Encode per line + EncodeBatch(vector<string_view>) of the C++ facade, device emulated: the reference's ids."""
    b = _build(emu=True)
    args = [b, os.path.join(fixtures.GOLDEN, model + ".model"), os.path.join(fixtures.GOLDEN, "botchan.txt")]
    if opts:
        args.append(opts)
    out = subprocess.run(args, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ids = np.array([int(x) for line in out.stdout.split("\n") for x in line.split()], dtype=np.int32)
    np.testing.assert_array_equal(ids, golden_arrays[key + "__ids"])


@pytest.mark.gpu
@pytest.mark.parametrize("model,opts,key", [("test_model", "", "test_model__botchan"),
                                             ("test_model", "bos:eos", "test_model__botchan__bos-eos"),
                                             ("bpe1k", "", "bpe1k__botchan")])
def test_facade_matches_golden(model, opts, key, golden_arrays, tmp_path):
    b = _build()
    args = [b, os.path.join(fixtures.GOLDEN, model + ".model"), os.path.join(fixtures.GOLDEN, "botchan.txt")]
    if opts:
        args.append(opts)
    out = subprocess.run(args, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ids = np.array([int(x) for line in out.stdout.split("\n") for x in line.split()], dtype=np.int32)
    np.testing.assert_array_equal(ids, golden_arrays[key + "__ids"])


@pytest.mark.gpu
@pytest.mark.parametrize("model,opts", [("test_model", ""), ("uni1k_bf", "bos:eos:unk_piece"), ("bpe1k", "reverse"), ("test_ja_model", "")])
def test_facade_pieces_and_normalize(model, opts, oracle, tmp_path):
    """Encode(input, SentencePieceText*), EncodeAsPieces and Normalize of the facade against the oracle."""
    b = _build()
    src = open(os.path.join(fixtures.GOLDEN, "botchan.txt"), "rb").read().split(b"\n")[:300]
    src += ["吾輩は猫である。 名前は まだ無い".encode(), b"  lead and trail  ", b"", b"\xff\xfe broken", "ＡＢＣ㍿".encode()]
    path = tmp_path / "in.txt"
    path.write_bytes(b"\n".join(src) + b"\n")
    out = subprocess.run([b, os.path.join(fixtures.GOLDEN, model + ".model"), str(path), opts, "--pieces"], capture_output=True)
    assert out.returncode == 0, out.stderr
    o = oracle.load(fixtures.model_blob(model))
    o.set_encode_extra_options(opts)
    offs = np.zeros(len(src) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in src])
    text = np.frombuffer(b"".join(src), dtype=np.uint8)
    ids, bg, en, io, pblob, poffs = o.encode_pieces(text, offs)
    norm, no, n2o = o.normalize_batch(text, offs)
    lines = out.stdout.decode().split("\n")
    assert len(lines) == len(src) + 1
    for i in range(len(src)):
        toks, rest = lines[i].split("N ")
        got = [t.split(":") for t in toks.split()]
        want = []
        for k in range(int(io[i]), int(io[i + 1])):
            pc = pblob[int(poffs[k]):int(poffs[k + 1])]
            want.append([pc.hex() or "-", str(int(ids[k])), str(int(bg[k])), str(int(en[k]))])
        assert got == want, i
        f = rest.split()
        nb = norm[int(no[i]):int(no[i + 1])].tobytes()
        assert f[0] == (nb.hex() or "-"), i
        a = n2o[int(no[i]) + i:int(no[i + 1]) + i + 1]
        want_a = [] if (len(a) == 1 and a[0] == 0xFFFFFFFF) else [str(int(v)) for v in a]
        assert f[1:] == want_a, i


def _status_case(tmp_path):
    """A BPE model with a one-character CONTROL piece: a sentence that holds it is the reference's kInternal "all
    normalized characters are not consumed." (sentencepiece_processor.cc:628)."""
    from sentencepiece import sentencepiece_model_pb2 as pb
    m = pb.ModelProto()
    m.ParseFromString(fixtures.model_blob("bpe1k"))
    p = m.pieces.add()
    p.piece, p.score, p.type = "\u2603", 0.0, 3
    mp = tmp_path / "ctl.model"
    mp.write_bytes(m.SerializeToString())
    lines = [b"hello world", "say \u2603 then".encode(), b"", b"the end", "\u2603".encode()]
    tp = tmp_path / "in.txt"
    tp.write_bytes(b"\n".join(lines) + b"\n")
    return str(mp), str(tp), [0, 13, 0, 0, 13]


def _check_status(out, want):
    assert out.returncode == 0, out.stderr
    rows = [r.split("\t") for r in out.stdout.split("\n") if r]
    assert len(rows) == len(want)
    for r, code in zip(rows, want):
        for col in r[:3]:                 # Encode(ids), Encode(spt), Encode(pieces): the sentence's own Status
            c, msg = col.split("|", 1)
            assert int(c) == code, rows
            assert msg == ("all normalized characters are not consumed." if code else ""), rows
        if code:
            assert r[3] == "0" and r[4] == "0"      # EncodeAsIds swallows the error and returns nothing


def test_facade_encode_returns_the_sentence_status_emulated(tmp_path):
    b = _build(emu=True)
    mp, tp, want = _status_case(tmp_path)
    _check_status(subprocess.run([b, mp, tp, "", "--status"], capture_output=True, text=True), want)


@pytest.mark.gpu
def test_facade_encode_returns_the_sentence_status(tmp_path):
    """The C++ drop-in's Encode returns what the reference returns for a failing sentence: code 13 and its message, no ids."""
    b = _build()
    mp, tp, want = _status_case(tmp_path)
    _check_status(subprocess.run([b, mp, tp, "", "--status"], capture_output=True, text=True), want)


def _extras_case(tmp_path):
    lines = ["I saw a girl with a telescope.", "  Hello   world  ", "吾輩は猫である。", "tab\there 　x", "a", "€uro ＡＢ ㍿", "the theatre"]
    tp = tmp_path / "in.txt"
    tp.write_bytes("\n".join(lines).encode() + b"\n")
    vp = tmp_path / "vocab.tsv"
    vp.write_bytes("▁I\t5\n▁saw\t1\n▁a\t9\ns\t3\ne\n▁the\t2\n".encode())
    return lines, str(tp), str(vp)


def _check_extras(out, sp, ref, lines, vocab_path, decode_opts, opts=""):
    """sp: the Python mirror over the same library (its EncodeAsSerializedProto is pinned byte for byte to the reference's
    SerializeAsString elsewhere); ref: the compiled reference (Decode(pieces), GetScore, SetVocabulary)."""
    assert out.returncode == 0, out.stderr
    rows = out.stdout.decode().split("\n")
    unk = sp.IdToPiece(sp.unk_id())
    ref.set_decode_extra_options(decode_opts)
    k = 0
    for li, line in enumerate(lines):
        pcs = sp.EncodeAsPieces(line)
        if li % 3 == 0:
            pcs.insert(len(pcs) // 2, "zzqq-not-a-piece")
        if li % 3 == 1:
            pcs.append("▁outside")
            pcs.insert(0, "")
        if li % 5 == 2:
            pcs.append(unk)
        want = ref.decode_pieces(pcs)
        assert rows[k] == "D " + (want.hex() or "-"), (li, rows[k])
        assert rows[k + 1] == "P " + sp.EncodeAsSerializedProto(line).hex(), li
        k += 2
    ref.set_decode_extra_options("")
    scores = rows[k].split()[1:]
    assert len(scores) == min(64, sp.GetPieceSize())
    for i, v in enumerate(scores):
        assert np.float32(float(v)) == np.float32(ref.get_score(i)), i
    blob = sp.serialized_model_proto()
    fnv = 1469598103934665603
    for c in blob:
        fnv = ((fnv ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert rows[k + 1] == "M %d %d" % (len(blob), fnv)
    # LoadVocabulary(file, 2): the tokens with frequency >= 2 (no second column: 1) -- against the reference's SetVocabulary
    keep = ["▁I", "▁a", "s", "▁the"]
    ref.set_encode_extra_options(opts)
    ref.set_vocabulary(keep)
    b = np.frombuffer(lines[0].encode(), dtype=np.uint8)
    restricted, _ = ref.encode_batch(b, np.array([0, len(b)], dtype=np.uint64))
    ref.reset_vocabulary()
    plain, _ = ref.encode_batch(b, np.array([0, len(b)], dtype=np.uint64))
    v, r = rows[k + 2][1:].split("|")
    assert [int(x) for x in v.split()] == list(restricted)
    assert [int(x) for x in r.split()] == list(plain)


EXTRAS = [("test_model", "", ""), ("uni1k_bf", "bos:eos", "reverse"), ("bpe1k", "reverse", "unk")]


@pytest.mark.parametrize("model,opts,dopts", EXTRAS)
def test_facade_extras_emulated(model, opts, dopts, tmp_path):
    """Decode(pieces) with pieces outside the vocabulary, EncodeAsSerializedProto, GetScore, serialized_model_proto,
    LoadVocabulary -- through a base-class pointer (the facade's methods are virtual, as the reference's are)."""
    from tests import emulib, refshim
    if not refshim.available():
        pytest.skip("the compiled reference (oracle/_ref) is not built")
    b = _build(emu=True)
    lines, tp, vp = _extras_case(tmp_path)
    out = subprocess.run([b, os.path.join(fixtures.GOLDEN, model + ".model"), tp, opts, "--extras", vp, dopts], capture_output=True)
    sp = emulib.EmuLib().load(fixtures.model_blob(model)).sp
    sp.SetEncodeExtraOptions(opts)
    _check_extras(out, sp, refshim.RefLib().load(fixtures.model_blob(model)), lines, vp, dopts, opts)


@pytest.mark.gpu
@pytest.mark.parametrize("model,opts,dopts", EXTRAS)
def test_facade_extras(model, opts, dopts, tmp_path):
    from sentencepiece_amd.processor import SentencePieceProcessor
    from tests import refshim
    if not refshim.available():
        pytest.skip("the compiled reference (oracle/_ref) is not built")
    b = _build()
    lines, tp, vp = _extras_case(tmp_path)
    out = subprocess.run([b, os.path.join(fixtures.GOLDEN, model + ".model"), tp, opts, "--extras", vp, dopts], capture_output=True)
    sp = SentencePieceProcessor(model_proto=fixtures.model_blob(model))
    sp.SetEncodeExtraOptions(opts)
    _check_extras(out, sp, refshim.RefLib().load(fixtures.model_blob(model)), lines, vp, dopts, opts)
