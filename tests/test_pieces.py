"""EncodeAsPieces / the SentencePieceText fields and the batch Normalize:

  oracle vs the compiled reference      pieces(i).piece() / id / begin / end, Normalize(input, &norm, &norm_to_orig)
  device kernels (emulator) vs oracle   normalize kernels; spans with normalized ranges + normalized text -> pieces
  GPU through the C ABI vs oracle       -m gpu
"""
import numpy as np
import pytest

from tests import fixtures, pieceslib
from tests.test_spans import BIG, MODELS, inputs, packed

OPTS = ["", "unk_piece:bos:eos", "reverse:unk"]


@pytest.fixture(scope="module")
def ref():
    from tests import refshim
    if not refshim.available():
        pytest.skip("oracle/_ref/libspm_ref.so not built")
    return refshim.RefLib()


@pytest.fixture(scope="module")
def emu():
    from tests import emulib
    return emulib.EmuLib()


def same(a, b, names, what):
    for x, y, nm in zip(a, b, names):
        if isinstance(x, bytes):
            assert x == y, "%s: %s" % (what, nm)
        else:
            np.testing.assert_array_equal(np.asarray(x).astype(np.int64), np.asarray(y).astype(np.int64), err_msg="%s: %s" % (what, nm))


PN = ("ids", "begin", "end", "id_offsets", "pieces", "piece_offsets")
NN = ("normalized", "norm_offsets", "norm_to_orig")


@pytest.mark.parametrize("model", MODELS + BIG)
def test_oracle_pieces_and_normalize_match_reference(model, oracle, ref, corpora):
    blob = fixtures.model_blob(model)
    o, r = oracle.load(blob), ref.load(blob)
    for opts in OPTS if model in ("test_model", "bpe1k", "uni1k_bf") else ["", "reverse:unk"]:
        o.set_encode_extra_options(opts)
        r.set_encode_extra_options(opts)
        for name, (text, offs) in inputs(corpora, big=model in BIG):
            same(o.encode_pieces(text, offs), r.encode_pieces(text, offs), PN, "%s %s [%s]" % (model, name, opts))
            if not opts:
                same(o.normalize_batch(text, offs), r.normalize_batch(text, offs), NN, "%s %s" % (model, name))


def literal_fn(o_types, unk_opt):
    def literal(t):
        return o_types[t] in (3, 6) or (unk_opt and o_types[t] == 2)
    return literal


def piece_types(model):
    """SentencePiece.Type per id, read from the ModelProto with the product's own parser-independent helper."""
    from sentencepiece import sentencepiece_model_pb2 as pb
    m = pb.ModelProto()
    m.ParseFromString(fixtures.model_blob(model))
    return [p.type for p in m.pieces], [p.piece.encode("utf-8") for p in m.pieces]


@pytest.mark.parametrize("model", MODELS + BIG)
def test_emu_normalize_and_pieces(model, emu, oracle, corpora):
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob), oracle.load(blob)
    types, names = piece_types(model)
    for opts in OPTS if model in ("test_model", "bpe1k", "uni1k_bf") else ["", "reverse:unk"]:
        h.set_encode_extra_options(opts)
        o.set_encode_extra_options(opts)
        for name, (text, offs) in inputs(corpora, big=model in BIG):
            norm, no, n2o = h.normalize_batch(text, offs)
            assert h.status == 0
            if not opts:
                same((norm, no, n2o), o.normalize_batch(text, offs), NN, "%s %s" % (model, name))
            ids, b, e, io, nb, ne = h.encode_spans(text, offs, norm_spans=True)
            assert h.status == 0
            blob_p, poffs = pieceslib.compose_pieces(ids, nb, ne, io, norm, no, lambda t: names[t],
                                                     literal_fn(types, "unk" in opts))
            same((ids, b, e, io, blob_p, poffs), o.encode_pieces(text, offs), PN, "%s %s [%s]" % (model, name, opts))


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS + BIG)
def test_gpu_pieces_and_normalize(model, oracle, corpora):
    from sentencepiece_amd.processor import SentencePieceProcessor
    blob = fixtures.model_blob(model)
    sp, o = SentencePieceProcessor(model_proto=blob), oracle.load(blob)
    for opts in OPTS if model in ("test_model", "bpe1k", "uni1k_bf") else ["", "reverse:unk"]:
        sp.SetEncodeExtraOptions(opts)
        o.set_encode_extra_options(opts)
        cases = list(inputs(corpora))
        if not opts:
            cases += [(nm + "_full", fixtures.head(*corpora[nm], k)) for nm, k in (("botchan", 10 ** 6), ("mixed2k", 2000))]
        for name, (text, offs) in cases:
            what = "%s %s [%s]" % (model, name, opts)
            if not opts:
                same(sp.NormalizePacked(text, offs, with_offsets=True), o.normalize_batch(text, offs), NN, what)
            n = len(offs) - 1
            tb = np.asarray(text).tobytes()
            lines = [tb[int(offs[i]):int(offs[i + 1])] for i in range(n)]
            rows = sp.EncodeAsSentencePieceText(lines)
            ids, b, e, io, pblob, poffs = o.encode_pieces(text, offs)
            k = 0
            for i, row in enumerate(rows):
                assert len(row) == int(io[i + 1]) - int(io[i]), what
                for piece, t, surface, pb_, pe_ in row:
                    assert (t, pb_, pe_) == (int(ids[k]), int(b[k]), int(e[k])), what
                    assert piece == pblob[int(poffs[k]):int(poffs[k + 1])], what
                    assert surface == lines[i][pb_:pe_], what
                    k += 1
            assert k == len(ids)


@pytest.mark.gpu
def test_gpu_normalize_api_shapes(oracle):
    """The reference's Python forms: Normalize(str), Normalize(list), with_offsets (python/src/sentencepiece/__init__.py:907-915)."""
    from sentencepiece_amd.processor import SentencePieceProcessor
    blob = fixtures.model_blob("test_model")
    sp, o = SentencePieceProcessor(model_proto=blob), oracle.load(blob)
    assert sp.Normalize("ＫＡＤＯＫＡＷＡ  ＡＢＣ ") == o.normalize("ＫＡＤＯＫＡＷＡ  ＡＢＣ ".encode()).decode()
    out = sp.Normalize(["", "   ", "㍿ x"], with_offsets=True)
    assert out[0] == ("", []) and out[1] == ("", [])
    s, a = out[2]
    assert s == o.normalize("㍿ x".encode()).decode() and len(a) == len(s.encode()) + 1
    assert sp.EncodeAsPieces("hello world") == [p.decode() for p, *_ in sp.EncodeAsSentencePieceText("hello world")]


@pytest.mark.gpu
def test_gpu_encode_out_type_str_matches_wheel():
    """encode(out_type=str, add_bos / add_eos / reverse / emit_unk_piece) against the installed sentencepiece wheel
    on a BPE model (the wheel's BPE path is identical to the reference's, SURVEY finding 4)."""
    spm = pytest.importorskip("sentencepiece")
    from sentencepiece_amd.processor import SentencePieceProcessor
    blob = fixtures.model_blob("bpe1k")
    sp = SentencePieceProcessor(model_proto=blob)
    ref = spm.SentencePieceProcessor(model_proto=blob)
    lines = ["Hello world.", "  two  spaces ", "吾輩は猫 cat", "", "I saw a girl with a telescope."]
    for kw in ({}, {"add_bos": True, "add_eos": True}, {"reverse": True, "add_eos": True}, {"emit_unk_piece": True},
               {"add_bos": True, "reverse": True, "emit_unk_piece": True}):
        assert sp.encode(lines, out_type=str, **kw) == ref.encode(lines, out_type=str, **kw), kw
        assert sp.encode(lines[2], out_type=str, **kw) == ref.encode(lines[2], out_type=str, **kw), kw


def _ref_serialized(r, text, offs):
    import ctypes as C
    text = np.ascontiguousarray(text, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    cap = int(len(text)) * 120 + 64 * n + 256
    out = np.empty(cap, dtype=np.uint8)
    oo = np.zeros(n + 1, dtype=np.uint64)
    fn = r.lib.spmref_encode_serialized_batch
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    tot = fn(r.h, text.ctypes.data if len(text) else None, offs.ctypes.data, n, out.ctypes.data, cap, oo.ctypes.data)
    assert tot >= 0
    b = out[:tot].tobytes()
    return [b[int(oo[i]):int(oo[i + 1])] for i in range(n)]


@pytest.mark.parametrize("model", ["test_model", "uni1k_bf", "bpe1k", "bpe1k_llama", "uni1k_uds", "test_ja_model"])
def test_emu_serialized_proto(model, emu, ref, corpora):
    """EncodeAsSerializedProto: the device outputs (emulated) through the host-side wire-format assembly
    (sentencepiece_amd/spt_proto.py) against the compiled reference's SerializeAsString, byte for byte."""
    from sentencepiece_amd import spt_proto
    blob = fixtures.model_blob(model)
    h, r = emu.load(blob), ref.load(blob)
    types, names = piece_types(model)
    for opts in ("", "reverse:bos:eos", "unk_piece"):
        h.set_encode_extra_options(opts)
        r.set_encode_extra_options(opts)
        rev = opts.split(":").count("reverse") % 2 == 1
        lit = literal_fn(types, "unk" in opts)
        for name, (text, offs) in inputs(corpora):
            want = _ref_serialized(r, text, offs)
            norm, no, _ = h.normalize_batch(text, offs)
            ids, b, e, io, nb, ne = h.encode_spans(text, offs, norm_spans=True)
            nbytes = norm.tobytes()
            tb = np.asarray(text).tobytes()
            for i in range(len(offs) - 1):
                raw = tb[int(offs[i]):int(offs[i + 1])]
                lo, hi = int(io[i]), int(io[i + 1])
                has = spt_proto.surface_flags(ids[lo:hi], nb[lo:hi], lambda t: types[t] == 6, lambda t: types[t] == 3, rev)
                pcs = []
                for k in range(lo, hi):
                    t = int(ids[k])
                    piece = names[t] if lit(t) else nbytes[int(no[i]) + int(nb[k]):int(no[i]) + int(ne[k])]
                    pcs.append((piece, t, raw[int(b[k]):int(e[k])] if has[k - lo] else None, int(b[k]), int(e[k])))
                assert spt_proto.serialize(raw, pcs) == want[i], (model, name, opts, i)


@pytest.mark.gpu
def test_gpu_serialized_proto(ref, corpora):
    from sentencepiece_amd.processor import SentencePieceProcessor
    for model in ("test_model", "uni1k_bf", "bpe1k"):
        blob = fixtures.model_blob(model)
        sp, r = SentencePieceProcessor(model_proto=blob), ref.load(blob)
        for opts in ("", "reverse:bos:eos"):
            sp.SetEncodeExtraOptions(opts)
            r.set_encode_extra_options(opts)
            for name, (text, offs) in inputs(corpora):
                tb = np.asarray(text).tobytes()
                lines = [tb[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
                assert sp.EncodeAsSerializedProto(lines) == _ref_serialized(r, text, offs), (model, name, opts)


def test_processor_glue_with_emulated_device(emu, ref, corpora):
    """The Python facade's own assembly code (EncodeAsSentencePieceText / EncodeAsPieces / EncodeAsSerializedProto /
    Normalize) with the device calls answered by the emulator: what runs on a GPU box, minus the GPU."""
    from sentencepiece_amd.processor import SentencePieceProcessor
    for model, opts in (("uni1k_bf", "reverse:bos:eos"), ("test_model", ""), ("bpe1k", "unk_piece")):
        blob = fixtures.model_blob(model)
        h, r = emu.load(blob), ref.load(blob)
        h.set_encode_extra_options(opts)
        r.set_encode_extra_options(opts)
        sp = h.sp                     # the product's Python facade, bound to the emulated library
        text, offs = next(x for nm, x in inputs(corpora) if nm == "extra")
        tb = np.asarray(text).tobytes()
        lines = [tb[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
        assert sp.EncodeAsSerializedProto(lines) == _ref_serialized(r, text, offs)
        assert sp.EncodeAsSerializedProto(lines[3]) == _ref_serialized(r, text, offs)[3]
        ids, b, e, io, pblob, poffs = r.encode_pieces(text, offs)
        got = sp.EncodeAsPieces(lines)
        k = 0
        for row in got:
            for p in row:
                assert p.encode("utf-8", "surrogateescape") == pblob[int(poffs[k]):int(poffs[k + 1])]
                k += 1
        assert k == len(ids)
        if not opts:
            s, a = sp.Normalize(lines[7].decode("utf-8", "surrogateescape"), with_offsets=True)
            nn, no, n2o = r.normalize_batch(*packed_one(lines[7]))
            assert s.encode("utf-8", "surrogateescape") == nn.tobytes() and a == [int(v) for v in n2o]


def packed_one(b):
    return np.frombuffer(b, dtype=np.uint8), np.array([0, len(b)], dtype=np.uint64)


def test_immutable_proto_glue(emu, ref, corpora):
    """encode(out_type="immutable_proto"): the character spans of the reference's Python wrapper
    (EncodeAsImmutableProto + ConvertToUnicodeSpans of the compiled reference), strings and SerializeAsString; device
    calls emulated."""
    import ctypes as C
    from sentencepiece_amd.processor import SentencePieceProcessor
    for model in ("bpe1k", "test_model", "uni1k_bf"):
        blob = fixtures.model_blob(model)
        h, r = emu.load(blob), ref.load(blob)
        sp = h.sp
        text, offs = next(x for nm, x in inputs(corpora) if nm == "extra")
        tb = np.asarray(text).tobytes()
        lines = [tb[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]

        def valid(x):       # (the reference's ConvertToUnicodeSpans writes past its table on a truncated last character)
            try:
                x.decode("utf-8")
                return True
            except UnicodeDecodeError:
                return False
        lines = [x for x in lines if valid(x)]
        text, offs = packed(lines)
        tb = np.asarray(text).tobytes()
        got = sp.EncodeAsImmutableProto(lines)
        n = len(lines)
        cap = len(tb) * 12 + 8 * n + 64
        b = np.zeros(cap, dtype=np.uint32)
        e = np.zeros(cap, dtype=np.uint32)
        io = np.zeros(n + 1, dtype=np.uint64)
        fn = r.lib.spmref_encode_unicode_spans_batch
        fn.restype = C.c_int64
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        offs64 = np.ascontiguousarray(offs, dtype=np.uint64)
        tot = fn(r.h, np.ascontiguousarray(text).ctypes.data, offs64.ctypes.data, n, b.ctypes.data, e.ctypes.data, cap, io.ctypes.data)
        assert tot >= 0
        ser = _ref_serialized(r, text, offs)
        for i, g in enumerate(got):
            lo, hi = int(io[i]), int(io[i + 1])
            assert [p.begin for p in g.pieces] == b[lo:hi].tolist(), (model, i, lines[i][:30])
            assert [p.end for p in g.pieces] == e[lo:hi].tolist(), (model, i, lines[i][:30])
            assert g.SerializeAsString() == ser[i]
