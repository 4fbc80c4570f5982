"""Known-answer tests transcribed from the reference's normalizer tests (src/normalizer_test.cc:37-147: NormalizeTest,
NormalizeWithoutDummyPrefixTest, NormalizeTreatWSAsSuffixTest, NormalizeWithoutRemoveExtraWhitespacesTest,
NormalizeWithoutEscapeWhitespacesTest) through the product's batch `Normalize` (device kernels under the emulator here):
the nmt_nfkc rules (the trainer's default) with the normalizer_spec / trainer_spec switches each test sets, against the reference's expected strings AND, where it is built,
the compiled reference's `Normalize` on the same ModelProto."""
import pytest

from tests import refshim

WS = "▁"


_SPEC = {}


def _nmt_nfkc_spec():
    """normalizer_spec (with its precompiled charsmap) of nmt_nfkc, the trainer's default rule -- what
    SentencePieceTrainer::GetNormalizerSpec("nmt_nfkc") gives the reference's tests -- from a throw-away model the pip wheel
    trains on the spot (the wheel is test tooling: fixtures and ModelProto editing only)."""
    if "spec" not in _SPEC:
        import io
        import sentencepiece as spm
        from sentencepiece import sentencepiece_model_pb2 as pb
        buf = io.BytesIO()
        spm.SentencePieceTrainer.train(sentence_iterator=iter(["abc def ghi jkl %d" % i for i in range(200)]), model_writer=buf,
                                       vocab_size=40, model_type="unigram", hard_vocab_limit=False)
        m = pb.ModelProto()
        m.ParseFromString(buf.getvalue())
        assert m.normalizer_spec.name == "nmt_nfkc" and len(m.normalizer_spec.precompiled_charsmap) > 1000
        _SPEC["spec"] = m.normalizer_spec
    return _SPEC["spec"]


def _model(add_dummy_prefix=True, remove_extra_ws=True, escape=True, suffix=False):
    """ModelProto = the nmt_nfkc normalizer_spec with the switches of the test, a minimal vocabulary."""
    from sentencepiece import sentencepiece_model_pb2 as pb
    m = pb.ModelProto()
    m.normalizer_spec.CopyFrom(_nmt_nfkc_spec())
    m.normalizer_spec.add_dummy_prefix = add_dummy_prefix
    m.normalizer_spec.remove_extra_whitespaces = remove_extra_ws
    m.normalizer_spec.escape_whitespaces = escape
    m.trainer_spec.treat_whitespace_as_suffix = suffix
    for piece, typ in (("<unk>", 2), ("<s>", 3), ("</s>", 3), ("a", 1), (WS, 1)):
        p = m.pieces.add()
        p.piece, p.score, p.type = piece, 0.0, typ
    return m.SerializeToString()


@pytest.fixture(scope="module", params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def emu(request):
    """The product's C ABI: on the CPU model of the wavefront (tests/emulib.py EmuLib), and -- the -m gpu twin of every test
    of this file -- libspmx.so on the device (GpuLib)."""
    from tests import emulib
    return emulib.backend(request.param)


def _check(emu, blob, kats):
    sp = emu.load(blob).sp
    ref = refshim.RefLib().load(blob) if refshim.available() else None
    got = sp.Normalize([k for k, _ in kats])
    for (inp, want), g in zip(kats, got):
        assert g == want, (inp, g, want)
        if ref is not None:
            assert ref.normalize(inp).decode("utf-8") == want, inp


def test_normalize(emu):
    kats = [("", ""), ("      ", ""), ("　", ""),
            ("ABC", WS + "ABC"), (" ABC ", WS + "ABC"), (" A  B  C ", WS + "A" + WS + "B" + WS + "C"), ("   ABC   ", WS + "ABC"),
            ("   ＡＢＣ   ", WS + "ABC"), ("　　ABC", WS + "ABC"), ("　　ABC　　", WS + "ABC"),
            ("①②③", WS + "123"),                       # NFKC char to char
            ("㍿", WS + "株式会社"),                     # NFKC char to multi-char
            (" ｸﾞｰｸﾞﾙ ", WS + "グーグル"),                # half-width katakana: composition happens
            (" I  saw a　 　girl　　", WS + "I" + WS + "saw" + WS + "a" + WS + "girl")]
    kats += [(chr(c), "") for c in (0x7F, 0x8F, 0x9F, 0x0B)] + [(chr(c), "") for c in range(0x10, 0x20)]   # control characters are removed
    _check(emu, _model(), kats)


def test_normalize_without_dummy_prefix(emu):
    kats = [("", ""), ("      ", ""), ("　", ""), ("ABC", "ABC"), (" ABC ", "ABC"), (" A  B  C ", "A" + WS + "B" + WS + "C"),
            ("   ABC   ", "ABC"), ("   ＡＢＣ   ", "ABC"), ("　　ABC", "ABC"), ("　　ABC　　", "ABC")]
    _check(emu, _model(add_dummy_prefix=False), kats)


def test_normalize_treat_whitespace_as_suffix(emu):
    kats = [("", ""), ("      ", ""), ("　", ""), ("ABC", "ABC" + WS), (" ABC ", "ABC" + WS),
            (" A  B  C ", "A" + WS + "B" + WS + "C" + WS), ("   ABC   ", "ABC" + WS)]
    _check(emu, _model(suffix=True), kats)


def test_normalize_without_remove_extra_whitespaces(emu):
    kats = [("", ""), ("      ", WS * 7), ("　", WS * 2), ("ABC", WS + "ABC"), (" ABC ", WS * 2 + "ABC" + WS),
            ("  A  B  C  ", WS * 3 + "A" + WS * 2 + "B" + WS * 2 + "C" + WS * 2)]
    _check(emu, _model(remove_extra_ws=False), kats)


def test_normalize_without_escape_whitespaces(emu):
    kats = [("", ""), ("      ", ""), ("　", ""), ("ABC", "ABC"), (" ABC ", "ABC"), ("  A  B  C  ", "A B C"), ("A　 B　 C", "A B C")]
    _check(emu, _model(add_dummy_prefix=False, remove_extra_ws=True, escape=False), kats)
