"""Known-answer tests transcribed from the reference's PROCESSOR tests (src/sentencepiece_processor_test.cc), the ones
that pin behaviour of this path beyond the model-level KATs of tests/test_reference_kats.py:

  SkipNormalizationTest        :1375-1405  a USER_DEFINED piece is copied verbatim past a case-folding normalizer
  ExtraOptionsUndefinedTest    :1407-1425  `bos` / `eos` options on a model without those pieces are an error
  OverrideSpecialPieceTest     :1427-1459  unk / bos / eos / pad ids follow trainer_spec's piece names
  EncodeTest / DecodeTest      :130-417, :544-709  (the extra-option forms: reverse, bos, eos and their orders)

through the emulated product library here (the C ABI + device bodies on the CPU), compared with the expected values of
the reference's tests AND, where the compiled reference is built, with what it returns for the same model bytes."""
import numpy as np
import pytest

from sentencepiece_amd import synth
from tests import refshim

UNK, CONTROL, USER_DEFINED = 2, 3, 4
WS = "▁"


def _model(pieces, trainer=b"", normalizer=b""):
    out = bytearray()
    for p, s, t in pieces:
        out += synth._piece_msg(p.encode("utf-8"), s, t)
    if trainer:
        out += b"\x12" + synth._varint(len(trainer)) + trainer
    if normalizer:
        out += b"\x1a" + synth._varint(len(normalizer)) + normalizer
    return bytes(out)


def _str_field(tag, s):
    b = s.encode("utf-8")
    return synth._varint(tag << 3 | 2) + synth._varint(len(b)) + b


@pytest.fixture(scope="module", params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def emu(request):
    """The product's C ABI: on the CPU model of the wavefront (tests/emulib.py EmuLib), and -- the -m gpu twin of every test
    of this file -- libspmx.so on the device (GpuLib)."""
    from tests import emulib
    return emulib.backend(request.param)


def _nmt_nfkc_cf_spec():
    """normalizer_spec (with its precompiled charsmap) of --normalization_rule_name=nmt_nfkc_cf: from a throw-away model the
    pip wheel trains on the spot (the wheel is test tooling only: fixtures and ModelProto editing)."""
    import io
    import sentencepiece as spm
    from sentencepiece import sentencepiece_model_pb2 as pb
    buf = io.BytesIO()
    lines = ["abc def ghi jkl %d" % i for i in range(200)]
    spm.SentencePieceTrainer.train(sentence_iterator=iter(lines), model_writer=buf, vocab_size=40, model_type="unigram",
                                   normalization_rule_name="nmt_nfkc_cf", hard_vocab_limit=False)
    m = pb.ModelProto()
    m.ParseFromString(buf.getvalue())
    assert m.normalizer_spec.name == "nmt_nfkc_cf" and len(m.normalizer_spec.precompiled_charsmap) > 1000
    return m.normalizer_spec.SerializeToString()


def test_skip_normalization_of_user_defined_symbols(emu):
    try:
        spec = _nmt_nfkc_cf_spec()
    except Exception as e:                      # (no such fixture here: nothing to case-fold with)
        pytest.skip("no nmt_nfkc_cf normalizer spec among the fixtures: %r" % (e,))
    pieces = [("<unk>", 0.0, UNK), ("<USER>", 0.0, USER_DEFINED)] + [(c, s, 1) for c, s in
              (("a", 0.0), ("b", 0.3), ("c", 0.2), ("u", 0.2), ("s", 0.2), ("e", 0.2), ("r", 0.2))]
    blob = _model(pieces, normalizer=spec)
    sp = emu.load(blob).sp
    want = [WS, "a", "b", "<USER>", "c", "<", "u", "s", "e", "r", ">"]
    assert sp.EncodeAsPieces("AB<USER>C<uSEr>") == want
    if refshim.available():
        ref = refshim.RefLib().load(blob)
        text, offs = synth.pack([b"AB<USER>C<uSEr>"])
        rids, rio = ref.encode_batch(text, offs)
        ids, io = sp.EncodePacked(text, offs)
        np.testing.assert_array_equal(ids, rids)


def test_extra_options_undefined(emu):
    blob = _model([("<unk>", 0.0, UNK), ("a", 0.0, 1), ("b", 0.3, 1), ("c", 0.2, 1), ("ab", 1.0, 1)])
    sp = emu.load(blob).sp
    with pytest.raises(RuntimeError, match="is not defined"):
        sp.SetEncodeExtraOptions("bos")
    with pytest.raises(RuntimeError, match="is not defined"):
        sp.SetDecodeExtraOptions("eos")
    sp.SetEncodeExtraOptions("reverse")        # (needs no piece)
    sp.SetEncodeExtraOptions("")
    with pytest.raises(RuntimeError):
        sp.SetEncodeExtraOptions("foo")         # "option "foo" is not available."  (:1058)
    assert sp.EncodeAsIds("ab") == sp.EncodeAsIds("ab")


def test_override_special_pieces(emu):
    trainer = _str_field(45, "__UNK__") + _str_field(46, "__BOS__") + _str_field(47, "__EOS__") + _str_field(48, "__PAD__")
    blob = _model([("__UNK__", 0.0, UNK), ("__BOS__", 0.0, CONTROL), ("__EOS__", 0.0, CONTROL), ("a", 0.0, 1), ("b", 0.3, 1)],
                  trainer=trainer)
    sp = emu.load(blob).sp
    assert (sp.unk_id(), sp.bos_id(), sp.eos_id(), sp.pad_id()) == (0, 1, 2, -1)
    assert [sp.IdToPiece(i) for i in (0, 1, 2)] == ["__UNK__", "__BOS__", "__EOS__"]
    sp.SetEncodeExtraOptions("bos:eos")
    ids = sp.EncodeAsIds("ab")
    assert ids[0] == 1 and ids[-1] == 2
    if refshim.available():
        ref = refshim.RefLib().load(blob)
        ref.set_encode_extra_options("bos:eos")
        text, offs = synth.pack([b"ab", b"", b"ba ab"])
        rids, rio = ref.encode_batch(text, offs)
        gids, gio = sp.EncodePacked(text, offs)
        np.testing.assert_array_equal(gio, rio)
        np.testing.assert_array_equal(gids, rids)


@pytest.mark.parametrize("opts", ["reverse", "bos", "eos", "bos:eos", "reverse:bos", "bos:reverse", "eos:reverse:bos", "reverse:eos:bos"])
def test_encode_extra_option_orders(opts, emu):
    """ApplyExtraOptions applies the options IN THE ORDER LISTED (src/sentencepiece_processor.cc:1019-1064): `bos:reverse` puts
    the bos at the END."""
    from tests import fixtures
    blob = fixtures.model_blob("test_model")
    sp = emu.load(blob).sp
    plain = sp.EncodeAsIds("hello world")
    ids = list(plain)
    for o in opts.split(":"):
        ids = ids[::-1] if o == "reverse" else ([sp.bos_id()] + ids if o == "bos" else ids + [sp.eos_id()])
    sp.SetEncodeExtraOptions(opts)
    assert sp.EncodeAsIds("hello world") == ids
    if refshim.available():
        ref = refshim.RefLib().load(blob)
        ref.set_encode_extra_options(opts)
        text, offs = synth.pack([b"hello world"])
        rids, _ = ref.encode_batch(text, offs)
        assert rids.tolist() == ids


# ---- Decode KATs (the reference tests use mock models: the same pieces as real ModelProtos here) ----

def _decode(sp, ids):
    text, offs = sp.DecodePacked(np.asarray(ids, dtype=np.int32), np.asarray([0, len(ids)], dtype=np.uint64))
    return bytes(np.asarray(text)[:int(offs[1])])


def _ref_decode(blob, ids):
    ref = refshim.RefLib().load(blob)
    text, offs = ref.decode_batch(np.asarray(ids, dtype=np.int32), np.asarray([0, len(ids)], dtype=np.uint64))
    return bytes(np.asarray(text)[:int(offs[1])])


@pytest.mark.parametrize("remove_extra_ws,want", [(0, b" ABC DEFG H"), (1, b"ABC DEFG H")])
def test_dummy_prefix_decode(remove_extra_ws, want, emu):
    """DummyPrefixDecodeTest (:711-789): <s> ▁ ▁ABC <unk> ▁DE F G▁H I </s> with unk_surface "" -- only ONE leading space symbol is
    the dummy prefix's; whether the second one's space stays depends on remove_extra_whitespaces.  (The reference's test decodes
    PIECES, where the unknown piece "I" is its own surface: " ABC DEFG HI"; ids carry no strings, so <unk> decodes to
    unk_surface, here "".)"""
    pieces = [("<unk>", 0.0, UNK), ("<s>", 0.0, CONTROL), ("</s>", 0.0, CONTROL), (WS + "ABC", 0.0, 1), (WS + "DE", 0.0, 1),
              ("F", 0.0, 1), ("G" + WS + "H", 0.0, 1), (WS, 0.0, 1)]
    trainer = synth._varint(44 << 3 | 2) + synth._varint(0)                     # unk_surface = ""
    norm = b"\x18\x01" + b"\x20" + bytes([remove_extra_ws]) + b"\x28\x01"          # add_dummy_prefix, remove_extra_whitespaces, escape
    blob = _model(pieces, trainer=trainer, normalizer=norm)
    ids = [1, 7, 3, 0, 4, 5, 6, 0, 2]                                          # ("I" is not a piece: PieceToId gives <unk>)
    sp = emu.load(blob).sp
    assert _decode(sp, ids) == want
    if refshim.available():
        assert _ref_decode(blob, ids) == want


def test_byte_fallback_decode(emu):
    """ByteFallbackDecodeTest (:791-951): byte pieces are reassembled into characters; bytes that are not valid UTF-8 become
    U+FFFD one by one; U+FFFD spelled in bytes stays U+FFFD."""
    pieces = [("<unk>", 0.0, UNK), ("<s>", 0.0, CONTROL), ("</s>", 0.0, CONTROL), ("A", 0.0, 1), ("B", 0.0, 1), ("C", 0.0, 1)]
    pieces += [("<0x%02X>" % b, 0.0, 6) for b in range(256)]
    trainer = synth._varint(35 << 3 | 0) + b"\x01"                              # byte_fallback = true
    blob = _model(pieces, trainer=trainer)
    B = lambda *bs: [6 + b for b in bs]
    ids = [1, 3, 4] + B(0xE3, 0x81, 0x82) + B(0x5A) + B(0xCE, 0xA9) + [5] + B(0xE0, 0x80) + B(0xE3, 0x81, 0x84) + B(0xEF, 0xBF, 0xBD)
    want = "ABあZΩC��い�".encode("utf-8")
    sp = emu.load(blob).sp
    assert _decode(sp, ids) == want
    if refshim.available():
        assert _ref_decode(blob, ids) == want


@pytest.mark.parametrize("extra,want", [([], ["A", "B", "C"]), ([("AB", 2.0)], ["AB", "C"]), ([("AB", 2.0), ("BC", 5.0)], ["A", "BC"]),
                                         ([("AB", 2.0), ("BC", 5.0), ("ABC", 10.0)], ["ABC"])])
def test_lattice_viterbi_kat(extra, want, emu):
    """LatticeTest.ViterbiTest (src/unigram_model_test.cc:195-212): nodes A, B, C at score 0, then AB 2.0, BC 5.0, ABC 10.0 added
    one by one -- as a unigram model's pieces, through both encoders (EncodeOptimized and, by the extra entry point, kOriginal)."""
    from tests.test_reference_kats import build_model
    pieces = [("A", 0.0, 1), ("B", 0.0, 1), ("C", 0.0, 1)] + [(p, s, 1) for p, s in extra]
    blob = build_model(1, pieces)
    sp = emu.load(blob).sp
    assert sp.EncodeAsPieces("ABC") == want
    ids = {p: 3 + i for i, (p, _, _) in enumerate(pieces)}
    assert sp.EncodeAsIds("ABC") == [ids[w] for w in want]
    if refshim.available():
        ref = refshim.RefLib().load(blob)
        assert list(ref.encode(b"ABC")) == [ids[w] for w in want]
        ref.set_encoder_original()
        assert list(ref.encode(b"ABC")) == [ids[w] for w in want]
