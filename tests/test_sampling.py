"""SampleEncode (subword regularization, BPE-dropout) and the kOriginal Viterbi encoder on the device kernels.

What can be bit-exact is: kOriginal against the compiled reference switched to EncoderVersion::kOriginal; nbest_size 0/1
and BPE alpha = 0 against Encode; BPE alpha >= 1 (every merge skipped -- deterministic) against the reference's
SampleEncode.  The draws themselves come from generators keyed by (seed, sentence) -- the reference's thread-local mt19937
stream is not reproducible across processes -- so, like the reference's own tests (unigram_model_test.cc:429-470,
bpe_model_test.cc:252-295), the DISTRIBUTION is what is pinned: against the closed form exp(alpha * score) / Z over the
enumerated segmentations, and against the reference's own SampleEncode frequencies."""
import collections
import math

import numpy as np
import pytest

from tests import fixtures
from tests.test_nbest import nbest, sentences

UNIGRAM = ["test_model", "test_ja_model", "uni1k", "uni1k_bf", "uni1k_uds", "uni1k_suffix"]


class _Eng:
    """A loader of device-path processors: the wave emulator (CPU suite) or the HIP library (-m gpu)."""

    def __init__(self, kind):
        self.kind = kind
        if kind == "emu":
            from tests import emulib
            self.lib = emulib.EmuLib()

    def load(self, blob):
        if self.kind == "emu":
            return self.lib.load(blob).sp
        from sentencepiece_amd.processor import SentencePieceProcessor
        return SentencePieceProcessor(model_proto=blob)


@pytest.fixture(scope="module", params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def emu(request):
    return _Eng(request.param)


@pytest.fixture(scope="module")
def ref():
    from tests import refshim
    if not refshim.available():
        pytest.skip("oracle/_ref/libspm_ref.so not built")
    return refshim.RefLib()


def rows(ids, io):
    io = io.astype(np.int64)
    return [ids[io[i]:io[i + 1]].tolist() for i in range(len(io) - 1)]


@pytest.mark.parametrize("model", UNIGRAM)
def test_original_encoder_matches_reference(model, emu, ref, corpora):
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h, r = emu.load(blob), ref.load(blob)
    r.set_encoder_original()
    sents = [s for s in sentences(corpora) if len(s) <= 250]
    for opts in ("", "reverse:bos:eos"):
        h.SetEncodeExtraOptions(opts)
        r.set_encode_extra_options(opts)
        got = rows(*h.EncodeOriginalPacked(*synth.pack(sents)))
        for s, g in zip(sents, got):
            assert g == r.encode(s).tolist(), (model, s[:40], opts)


@pytest.mark.parametrize("model", ["test_model", "uni1k_bf", "bpe1k", "bpe1k_bf_uds", "bpe1k_noesc"])
def test_degenerate_sampling_is_encode(model, emu, corpora):
    """Unigram: nbest_size 0 / 1 is the plain encoder (src/sentencepiece_processor.cc:694-697).  BPE: every nbest_size
    is BPE-dropout (:688-693, IsNBestEncodeAvailable() is false), which is the plain encoder at alpha = 0."""
    from sentencepiece_amd import synth
    h = emu.load(fixtures.model_blob(model))
    sents = [s for s in sentences(corpora) if len(s) <= 250]
    text, offs = synth.pack(sents)
    want = rows(*h.EncodePacked(text, offs))
    if model.startswith("bpe"):
        for nb in (-1, 0, 1, 5):
            assert rows(*h.SampleEncodePacked(text, offs, nb, 0.0, seed=7)) == want
    else:
        for nb in (0, 1):
            assert rows(*h.SampleEncodePacked(text, offs, nb, 0.5, seed=7)) == want


@pytest.mark.parametrize("model", ["bpe1k", "bpe1k_noesc"])
def test_bpe_sample_encode_ignores_nbest_size(model, emu, ref, corpora):
    """A BPE model has no n-best: SampleEncode sends every nbest_size to BPE-dropout with alpha
    (src/sentencepiece_processor.cc:688-693).  alpha = 1 skips every merge, so it is bit-comparable with the reference's
    own SampleEncode at nbest_size 0, 1 and 5; nbest_size > 512 is the reference's error (:684)."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h, r = emu.load(blob), ref.load(blob)
    sents = [s for s in sentences(corpora) if len(s) <= 250][:60]
    text, offs = synth.pack(sents)
    for nb in (0, 1, 5):
        got = rows(*h.SampleEncodePacked(text, offs, nb, 1.0, seed=3))
        for s, g in zip(sents, got):
            assert g == r.sample_encode(s, nb, 1.0), (model, nb, s[:40])
    with pytest.raises(RuntimeError, match="nbest_size must be nbest_size <= 512"):
        h.SampleEncodePacked(text, offs, 513, 0.5, seed=1)


def test_unigram_sample_encode_rejects_large_nbest(emu, corpora):
    from sentencepiece_amd import synth
    h = emu.load(fixtures.model_blob("test_model"))
    with pytest.raises(RuntimeError, match="nbest_size must be nbest_size <= 512"):
        h.SampleEncodePacked(*synth.pack([b"hello world"]), 513, 0.5, seed=1)


@pytest.mark.parametrize("model", ["bpe1k", "bpe1k_bf_uds", "bpe1k_noesc", "bpe1k_llama"])
def test_bpe_dropout_alpha_one_matches_reference(model, emu, ref, corpora):
    """alpha >= 1 skips every merge (src/bpe_model.cc:131-135): deterministic, so bit-comparable."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h, r = emu.load(blob), ref.load(blob)
    sents = [s for s in sentences(corpora) if len(s) <= 250]
    got = rows(*h.SampleEncodePacked(*synth.pack(sents), -1, 1.0, seed=3))
    for s, g in zip(sents, got):
        assert g == r.sample_encode(s, -1, 1.0), (model, s[:40])


def test_bpe_dropout_distribution(emu, ref):
    """bpe_model_test.cc:252-295: one segmentation at alpha = 0, several at alpha > 0; and the frequencies agree with
    the reference's own sampler."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob("bpe1k")
    h, r = emu.load(blob), ref.load(blob)
    s = b"international understanding"
    n = 4000
    text, offs = synth.pack([s] * n)
    for alpha in (0.0, 0.1, 0.5):
        got = collections.Counter(tuple(x) for x in rows(*h.SampleEncodePacked(text, offs, -1, alpha, seed=11)))
        want = collections.Counter(tuple(r.sample_encode(s, -1, alpha)) for _ in range(n))
        if alpha == 0.0:
            assert len(got) == 1 and got.keys() == want.keys()
            continue
        assert len(got) > 1
        for k in set(got) | set(want):
            assert abs(got[k] - want[k]) / n < 0.03, (alpha, k, got[k], want[k])
    # different seeds draw differently, the same seed reproduces
    a = rows(*h.SampleEncodePacked(text[:int(offs[64])], offs[:65], -1, 0.5, seed=1))
    assert a == rows(*h.SampleEncodePacked(text[:int(offs[64])], offs[:65], -1, 0.5, seed=1))
    assert a != rows(*h.SampleEncodePacked(text[:int(offs[64])], offs[:65], -1, 0.5, seed=2))


@pytest.mark.parametrize("model,sent", [("test_model", b"hello world"), ("uni1k_bf", "café ab".encode()),
                                        ("test_ja_model", "東京都に行く".encode())])
def test_unigram_sample_distribution(model, sent, emu, oracle, ref):
    """Lattice::Sample draws a segmentation with probability exp(theta * score) / Z (src/unigram_model.cc:511-542;
    unigram_model_test.cc:429-470 holds it to 0.02 of the closed form)."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h, o, r = emu.load(blob), oracle.load(blob), ref.load(blob)
    npaths, paths, scores = nbest(o.lib.oracle_nbest_encode, o.h, sent, 1000)
    assert 1 < npaths < 1000                      # the whole lattice enumerated
    n = 6000
    text, offs = synth.pack([sent] * n)
    for alpha in (0.0, 0.2, 1.0):
        z = [math.exp(alpha * float(sc)) for sc in scores]
        prob = {tuple(p): v / sum(z) for p, v in zip(paths, z)}
        got = collections.Counter(tuple(x) for x in rows(*h.SampleEncodePacked(text, offs, -1, alpha, seed=5)))
        assert set(got) <= set(prob)
        for k, p in prob.items():
            assert abs(got[k] / n - p) < 0.02, (model, alpha, k, got[k] / n, p)
    # and the reference's sampler sits in the same place
    want = collections.Counter(tuple(r.sample_encode(sent, -1, 0.2)) for _ in range(n))
    got = collections.Counter(tuple(x) for x in rows(*h.SampleEncodePacked(text, offs, -1, 0.2, seed=9)))
    for k in set(got) | set(want):
        assert abs(got[k] - want[k]) / n < 0.03


def test_unigram_nbest_sampling(emu, oracle):
    """nbest_size > 1: one of the n best with probability ~ exp(alpha * score) (src/sentencepiece_processor.cc:700-716)."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob("test_model")
    h, o = emu.load(blob), oracle.load(blob)
    sent = b"hello world"
    k, paths, scores = nbest(o.lib.oracle_nbest_encode, o.h, sent, 4)
    n = 4000
    text, offs = synth.pack([sent] * n)
    for alpha in (0.0, 0.5):
        z = [math.exp(alpha * float(sc)) for sc in scores]
        got = collections.Counter(tuple(x) for x in rows(*h.SampleEncodePacked(text, offs, 4, alpha, seed=2)))
        assert set(got) <= {tuple(p) for p in paths}
        for p, v in zip(paths, z):
            assert abs(got[tuple(p)] / n - v / sum(z)) < 0.03


def test_python_encode_sampling_switch(emu):
    h = emu.load(fixtures.model_blob("test_model"))
    a = [tuple(h.Encode("hello world", enable_sampling=True, nbest_size=-1, alpha=0.1)) for _ in range(40)]
    assert len(set(a)) > 1
    assert h.Encode("hello world", enable_sampling=True, nbest_size=1, alpha=0.1) == h.Encode("hello world")


def test_no_length_limit(emu, ref, oracle):
    """Sentences beyond the first lattice launch's capacities (1024 normalized bytes) take the wide launch."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob("test_model")
    h, r = emu.load(blob), ref.load(blob)
    r.set_encoder_original()
    sents = [(b"the quick brown fox jumps over the lazy dog " * 700)[:30000], b"short one", b"x" * 5000,
             ("\u5409\u7965 caf\u00e9 " * 400).encode()]
    text, offs = synth.pack(sents)
    got = rows(*h.EncodeOriginalPacked(text, offs))
    for s, g in zip(sents, got):
        assert g == r.encode(s).tolist()
    # a sampled segmentation of a long sentence is still a segmentation of it: it decodes to the same text
    ids, io = h.SampleEncodePacked(text, offs, -1, 0.3, seed=4)
    dt, do = h.DecodePacked(ids, io)
    et, eo = h.DecodePacked(*h.EncodePacked(text, offs))
    np.testing.assert_array_equal(do, eo)
    np.testing.assert_array_equal(dt, et)
    assert rows(ids, io) != rows(*h.EncodePacked(text, offs))
