"""EncodeAsPieces / the SentencePieceText fields and the batch Normalize:

  oracle vs the compiled reference      pieces(i).piece() / id / begin / end, Normalize(input, &norm, &norm_to_orig)
  device kernels (emulator) vs oracle   normalize kernels; spans with normalized ranges + normalized text -> pieces
  GPU through the C ABI vs oracle       -m gpu
"""
import numpy as np
import pytest

from tests import fixtures, pieceslib
from tests.test_spans import BIG, MODELS, inputs

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
            norm, no, n2o = h.normalize_batch(text, offs, grid=2)
            assert h.status == 0
            if not opts:
                same((norm, no, n2o), o.normalize_batch(text, offs), NN, "%s %s" % (model, name))
            ids, b, e, io, nb, ne = h.encode_spans(text, offs, grid=2, norm_spans=True)
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
