"""Parity tests proper: the HIP path, called through the C ABI, against the
golden ids produced by the compiled reference (tests/golden/) and against the
oracle on the same inputs.  Bit-exact: these are integer ids."""
import numpy as np
import pytest

from tests import fixtures

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def procs():
    from sentencepiece_amd.processor import SentencePieceProcessor
    cache = {}

    def get(model):
        if model not in cache:
            cache[model] = SentencePieceProcessor(model_proto=fixtures.model_blob(model))
        return cache[model]
    return get


def _keys():
    import json
    import os
    with open(os.path.join(fixtures.GOLDEN, "manifest.json")) as f:
        return sorted(k for k in json.load(f) if not k.startswith("_"))


@pytest.mark.parametrize("key", _keys())
def test_golden(key, manifest, golden_arrays, corpora, procs, oracle):
    m = manifest[key]
    sp = procs(m["model"])
    sp.SetEncodeExtraOptions(m["options"])
    text, offs = corpora[m["corpus"]]
    ids, io = sp.EncodePacked(text, offs)
    sp.SetEncodeExtraOptions("")
    cnt = np.diff(io.astype(np.int64))
    gold_cnt = golden_arrays[key + "__cnt"].astype(np.int64)
    bad = np.flatnonzero(cnt != gold_cnt)
    assert bad.size == 0, "sentence %d: %d ids, reference %d" % (bad[0], cnt[bad[0]], gold_cnt[bad[0]])
    if key + "__ids" in golden_arrays:
        np.testing.assert_array_equal(ids, golden_arrays[key + "__ids"])
    assert len(ids) == m["tokens"]
    assert fixtures.sha(ids) == m["sha256"]
    # and the oracle on the same input, element by element
    o = oracle.load(fixtures.model_blob(m["model"]))
    if m["options"]:
        o.set_encode_extra_options(m["options"])
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)


def test_document_length_sentences(procs, oracle, corpora):
    """Inputs far beyond the staged length classes (SentencePiece is routinely fed whole documents): 100 KB of ASCII,
    60 KB of Japanese, 1 MiB of one character; a model the per-lane normalizers cannot take stops at 8192 bytes."""
    from sentencepiece_amd import synth
    bot, boffs = corpora["botchan"]
    ja, joffs = corpora["ja"]
    docs = [bot[:int(boffs[1600])].tobytes().replace(b"\n", b" "), ja[:int(joffs[300])].tobytes()[:60000 // 3 * 3],
            b"ab " * 349525, "短い".encode()]
    assert len(docs[0]) > 90000 and len(docs[2]) > 1000000
    text, offs = synth.pack(docs)
    for model in ("test_model", "test_ja_model", "uni32k"):
        sp = procs(model)
        ids, io = sp.EncodePacked(text, offs)
        oids, oio = oracle.load(fixtures.model_blob(model)).encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
    with pytest.raises(Exception) as ei:
        procs("uni1k_uds").EncodePacked(*synth.pack([b"y" * 9000]))
    assert "8192" in str(ei.value)
    with pytest.raises(Exception):
        procs("test_model").EncodePacked(*synth.pack([b"z" * (1 << 20) + b"!"]))


def bpe_documents():
    """Documents of 4-20 KB with URL-like words (20-200 characters), blobs of 300-1500 characters, CJK and accented runs."""
    from sentencepiece_amd import synth
    rng = np.random.default_rng(11)
    words = [b"hello", b"world", b"the", b"tokenizer", b"a", b"of"]
    al = b"abcdefghijklmnopqrstuvwxyz0123456789/_-.%"

    def longword(n):
        return bytes(al[int(k)] for k in rng.integers(0, len(al), size=n))
    docs = []
    for i in range(24):
        parts, ln, target = [], 0, int(rng.integers(4200, 20000))
        while ln < target:
            r = rng.random()
            if r < 0.03:
                w = b"https://" + longword(int(rng.integers(20, 200)))
            elif r < 0.035:
                w = longword(int(rng.integers(300, 1500)))
            elif r < 0.05:
                w = "日本語のテキスト".encode()
            elif r < 0.06:
                w = ("é" * int(rng.integers(1, 40))).encode()
            else:
                w = words[int(rng.integers(0, len(words)))]
            parts.append(w)
            ln += len(w) + 1
        docs.append(b" ".join(parts))
    return synth.pack(docs)


def test_bpe_document_length_sentences(procs, oracle, corpora):
    """BPE models take documents too (the lane form works word by word; a word that outgrows its LDS slots is merged in
    HBM): ids equal to the oracle's; a word of more than 4096 characters, and models that cannot be segmented word by
    word, stop with OUT_OF_RANGE."""
    from sentencepiece_amd import synth
    text, offs = bpe_documents()
    bot, boffs = corpora["botchan"]
    big = synth.pack([bot[:int(boffs[1600])].tobytes().replace(b"\n", b" "), b"ab " * 300000])
    for model in ("bpe1k", "bpe32k", "bpe1k_llama"):
        sp = procs(model)
        o = oracle.load(fixtures.model_blob(model))
        for t, of in ((text, offs), big):
            ids, io = sp.EncodePacked(t, of)
            oids, oio = o.encode_batch(t, of)
            np.testing.assert_array_equal(io, oio)
            np.testing.assert_array_equal(ids, oids)
    with pytest.raises(Exception):
        procs("bpe1k").EncodePacked(*synth.pack([b"x" * 5000]))
    with pytest.raises(Exception) as ei:
        procs("bpe1k_bf_uds").EncodePacked(*synth.pack([b"hello world " * 500]))
    assert "4096" in str(ei.value)
