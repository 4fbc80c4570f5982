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
    60 KB of Japanese, 1 MiB of one character, something longer than the last class."""
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
    t, of = synth.pack([b"y" * 9000, b"z" * (1 << 20) + b"!"])
    for model in ("uni1k_uds", "test_model"):
        ids, io = procs(model).EncodePacked(t, of)
        oids, oio = oracle.load(fixtures.model_blob(model)).encode_batch(t, of)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


def limit_documents(corpora, size):
    """A document of `size` bytes of English, one of Japanese, a word of size / 8 characters, NFKC expansions (U+FDFA ->
    18 characters each), malformed bytes (each becomes U+FFFD), leading whitespace, nothing, a short sentence."""
    from sentencepiece_amd import synth
    bot, _ = corpora["botchan"]
    ja, _ = corpora["ja"]
    reps = size // len(bot) + 1
    eng = np.tile(bot, reps)[:size].tobytes()
    jap = np.tile(ja, size // len(ja) + 1)[:size * 2 // 3].tobytes()
    docs = [eng, jap, b"x" * (size // 8) + b" " + b"0123456789" * (size // 40), bot[:300].tobytes(), b"",
            ("\ufdfa" * (size // 30)).encode(), b"\xff\xfe" * (size // 20), (" " * (size // 10) + "a").encode()]
    return synth.pack(docs)


@pytest.mark.parametrize("model", ["uni1k_uds", "uni1k_suffix", "bpe1k_noesc", "bpe1k_bf_uds"])
def test_no_length_limit(model, procs, oracle, corpora):
    """a 100 KB and a 1 MiB document through the models the fast forms take only in part
    (user-defined symbols, whitespace as suffix, BPE pieces that span words): bit-equal to the reference's algorithm,
    every status byte zero."""
    sp = procs(model)
    o = oracle.load(fixtures.model_blob(model))
    for size in (100_000, 1 << 20):
        text, offs = limit_documents(corpora, size)
        ids, io, st, failed = sp.EncodePackedEx(text, offs)
        assert failed == 0 and not st.any()
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


def test_spans_and_normalize_of_long_documents(procs, oracle, corpora):
    """The spans form and Normalize on 100 KB documents (beyond every staged class): lane-per-sentence kernels."""
    text, offs = limit_documents(corpora, 100_000)
    for model in ("test_model", "uni1k_uds", "bpe1k", "bpe1k_noesc"):
        sp = procs(model)
        o = oracle.load(fixtures.model_blob(model))
        got = sp.EncodeSpansPacked(text, offs)
        want = o.encode_spans(text, offs)
        for a, b, nm in zip(got, want, ("ids", "begin", "end", "id_offsets")):
            np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64), err_msg="%s %s" % (model, nm))
        gn = sp.NormalizePacked(text, offs, with_offsets=True)
        wn = o.normalize_batch(text, offs)
        for a, b, nm in zip(gn, wn, ("normalized", "norm_offsets", "norm_to_orig")):
            np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64), err_msg="%s %s" % (model, nm))


def test_failing_sentence_does_not_fail_the_batch(procs, oracle, corpora):
    """A sentence the reference's Encode fails (a one-character CONTROL piece among BPE symbols: "all normalized
    characters are not consumed", sentencepiece_processor.cc:628) yields no ids and status 13; the rest of the batch
    is encoded."""
    from sentencepiece import sentencepiece_model_pb2 as pb
    from sentencepiece_amd import synth
    from sentencepiece_amd.processor import SentencePieceProcessor
    m = pb.ModelProto()
    m.ParseFromString(fixtures.model_blob("bpe1k"))
    p = m.pieces.add()
    p.piece, p.score, p.type = "\u2603", 0.0, 3
    sp = SentencePieceProcessor(model_proto=m.SerializeToString())
    bot, boffs = corpora["botchan"]
    good = [bot[int(boffs[i]):int(boffs[i + 1])].tobytes() for i in range(200)]
    bad = "say \u2603 then".encode()
    text, offs = synth.pack(good[:77] + [bad] + good[77:])
    ids, io, st, failed = sp.EncodePackedEx(text, offs)
    oids, oio = oracle.load(fixtures.model_blob("bpe1k")).encode_batch(*synth.pack(good))
    assert failed == 1 and st[77] == 13 and int(st.sum()) == 13
    np.testing.assert_array_equal(ids, oids)
    np.testing.assert_array_equal(np.delete(np.diff(io.astype(np.int64)), 77), np.diff(oio.astype(np.int64)))


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
    """BPE models take documents too (the lane form works word by word; a sentence with a word that outgrows its LDS
    slots, and models that cannot be segmented word by word, take the long form): ids equal to the oracle's."""
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
    t, of = synth.pack([b"x" * 5000, b"hello world " * 500])
    for model in ("bpe1k", "bpe1k_bf_uds"):
        ids, io = procs(model).EncodePacked(t, of)
        oids, oio = oracle.load(fixtures.model_blob(model)).encode_batch(t, of)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["test_model", "bpe1k", "uni1k_uds"])
def test_text_at_any_alignment(model, procs, oracle, corpora):
    """The text pointer and offsets[0] are used as given: the kernels' 16-byte loads are aligned on the ABSOLUTE address
    (csrc/kernels_normlane.h), so a buffer at an odd address, a first offset that is not a multiple of 16 and a buffer
    that ends exactly where its allocation ends all encode like the aligned case."""
    import torch
    sp = procs(model)
    text, offs = fixtures.head(*corpora["botchan"], 900)
    want, wio = oracle.load(fixtures.model_blob(model)).encode_batch(text, offs)
    dev = torch.device("cuda", 0)
    for shift in (1, 5, 7, 15, 16, 33):
        # host form: `shift` bytes of other data before the first sentence, offsets not rebased
        pad = np.concatenate([np.full(shift, 0xE3, dtype=np.uint8), text])
        ids, io = sp.EncodePacked(pad, offs + np.uint64(shift))
        np.testing.assert_array_equal(io, wio)
        np.testing.assert_array_equal(ids, want)
        # device form: a tensor view at an odd address whose last byte is the last byte of the allocation
        buf = torch.empty(shift + len(text), dtype=torch.uint8, device=dev)
        buf[shift:] = torch.from_numpy(text).to(dev)
        d_ids, d_io, total = sp.EncodeDevice(buf[shift:], torch.from_numpy(offs.view(np.int64)).to(dev))
        np.testing.assert_array_equal(d_io.cpu().numpy().astype(np.uint64), wio)
        np.testing.assert_array_equal(d_ids[:total].cpu().numpy(), want)
        # spans + normalize forms read the same text
        got = sp.EncodeSpansPacked(pad, offs + np.uint64(shift))
        np.testing.assert_array_equal(np.asarray(got[0]), want)


def _load_with_env(model, env):
    """A processor whose handle reads the SPMX_* switches in `env` at load (restored afterwards)."""
    import os
    from sentencepiece_amd.processor import SentencePieceProcessor
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return SentencePieceProcessor(model_proto=fixtures.model_blob(model))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _escalation_inputs(corpora):
    from sentencepiece_amd import synth
    bot, boffs = corpora["botchan"]
    docs = [bot[:int(boffs[400])].tobytes().replace(b"\n", b" "), b"ab " * 3000, b"x" * 2500 + b" 0123456789", b"", b"short one",
            "日本語のテキスト ".encode() * 40, b"  lead and trail  "]
    return synth.pack(docs), fixtures.head(*corpora["botchan"], 1200)


@pytest.mark.parametrize("model", ["test_model", "uni1k_uds", "uni32k", "bpe1k", "bpe1k_noesc", "bpe1k_bf_uds", "bpe32k"])
def test_arena_overflow_and_retry_on_the_gpu(model, oracle, corpora):
    """The GPU twin of tests/test_emu.py::test_emu_arena_overflow_and_retry: SPMX_ARENA_FIRST caps the first attempt's id
    arena, so every kernel family (lane-per-sentence, sentence-per-wave, long form, the word path) overflows, reports it
    BEFORE the compaction runs, and the batch is encoded again with the arena the first attempt asked for."""
    sp = _load_with_env(model, {"SPMX_ARENA_FIRST": "600"})
    o = oracle.load(fixtures.model_blob(model))
    for text, offs in _escalation_inputs(corpora):
        ids, io, st, failed = sp.EncodePackedEx(text, offs)
        assert failed == 0 and not st.any()
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
    text, offs = fixtures.head(*corpora["botchan"], 300)
    got = sp.EncodeSpansPacked(text, offs)
    want = o.encode_spans(text, offs)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64))


@pytest.mark.parametrize("model", ["test_model", "uni1k_uds", "uni1k_suffix", "uni32k", "test_ja_model", "bpe1k", "bpe1k_noesc",
                                   "bpe1k_bf_uds", "bpe1k_llama", "bpe32k"])
def test_small_class_table_runs_every_escalation_list_on_the_gpu(model, oracle, corpora):
    """SPMX_CLASSES with a tiny first class (the table the CPU suite uses under the emulator): short inputs overflow their
    text columns, escalate class by class, take the overflow launch and the long form -- on hardware."""
    from tests.emulib import SMALL_CLASSES
    sp = _load_with_env(model, {"SPMX_CLASSES": SMALL_CLASSES})
    o = oracle.load(fixtures.model_blob(model))
    inputs = list(_escalation_inputs(corpora)) + [fixtures.head(*corpora["ja"], 400), corpora["edge"], fixtures.head(*corpora["mixed2k"], 300)]
    for text, offs in inputs:
        ids, io, st, failed = sp.EncodePackedEx(text, offs)
        assert failed == 0 and not st.any()
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
    text, offs = fixtures.head(*corpora["botchan"], 300)
    got = sp.EncodeSpansPacked(text, offs)
    want = o.encode_spans(text, offs)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64))
    gn = sp.NormalizePacked(text, offs, with_offsets=True)
    wn = o.normalize_batch(text, offs)
    for a, b in zip(gn, wn):
        np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64))
