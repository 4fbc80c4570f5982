"""The product's launch sequence (csrc/api.cc), device bodies (csrc/kernels*.h) and host table compiler run on the
CPU -- the C ABI of tests/emu/libspmx_emu.so, see tests/emulib.py -- and are compared with the oracle.  Small inputs
only (the model executes 64 fibers per wave); the full-size parity runs are the -m gpu tests."""
import numpy as np
import pytest

from tests import fixtures

MODELS = ["test_model", "test_ja_model", "uni1k", "bpe1k", "uni1k_bf", "bpe1k_bf_uds", "uni1k_uds",
          "uni1k_ident", "uni1k_suffix", "bpe1k_noesc", "uni32k", "bpe32k", "c5_250k", "c5_250k_bf", "bpe1k_llama"]


@pytest.fixture(scope="module")
def emu():
    from tests import emulib
    return emulib.EmuLib()


@pytest.mark.parametrize("model", MODELS)
def test_emu_edge_and_samples(model, emu, oracle, corpora):
    blob = fixtures.model_blob(model)
    h = emu.load(blob)
    o = oracle.load(blob)
    for name, k in (("edge", 10 ** 6), ("botchan", 120), ("mixed2k", 25)):
        text, offs = fixtures.head(*corpora[name], k)
        ids, io = h.encode_batch(text, offs)
        assert h.status == 0
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


K_COMPRESS = 1 << 9   # dev.h kNfCompressSp


@pytest.mark.parametrize("model", ["c5_250k", "test_ja_model", "uni1k_bf", "uni1k_uds"])
def test_emu_long_sentences(model, emu, oracle, corpora):
    """The long length classes (up to 4096 B raw) through the streaming kernels: the longest sentences of the
    mixed-script power-law corpus, the Japanese sample and the long edge cases."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h = emu.load(blob)
    o = oracle.load(blob)
    t, of = corpora["mixed2k"]
    n = len(of) - 1
    pick = np.concatenate([np.arange(n - 5, n), np.arange(n - 400, n - 5, 80)])
    text, offs = synth.gather_packed(t, of, pick)
    for tx, ox in ((text, offs), fixtures.head(*corpora["ja"], 12)):
        ids, io = h.encode_batch(tx, ox)
        assert h.status == 0
        oids, oio = o.encode_batch(tx, ox)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["c5_250k", "test_ja_model", "uni1k_bf", "bpe32k", "uni1k_ident", "bpe1k_llama"])
@pytest.mark.parametrize("mode", ["always", "never", "auto"])
def test_emu_normalizer_forms(model, mode, emu, oracle, corpora):
    """The three lane normalizers of the streaming kernels give the reference's text: the byte-stepping form that
    applies charsmap rules itself (fast_norm_stream), the character-stepping form for tiles of mostly non-ASCII text
    (char_norm_stream; SPMX_CHAR_NORM_ALWAYS=1 sends every tile there, SPMX_NO_CHAR_NORM=1 none), and norm_lane_any
    for what both give up.  Inputs: mixed-script text with real NFKC rules (full-width letters, U+3000, circled
    digits, U+337F -> four ideographs), kana with combining marks, ASCII letters followed by combining marks (a key that
    starts at the ASCII byte), rule results that are or begin with a space next to real spaces, malformed bytes."""
    from sentencepiece_amd import synth
    from tests import test_fuzz
    blob = fixtures.model_blob(model)
    env = {"always": {"SPMX_CHAR_NORM_ALWAYS": "1"}, "never": {"SPMX_NO_CHAR_NORM": "1"}, "auto": {}}[mode]
    env["SPMX_NO_WORD_KERNEL"] = "1"          # (every sentence through the streaming kernels)
    o = oracle.load(blob)
    mt, mo = synth.mixed_corpus(160, seed=11, hi=900)
    ft, fo = test_fuzz.fuzz_corpus(260, 21, corpora)
    hand = ["Ａ　Ｂ", "a\u00a0\u00a0b", " ´x", "´ ´", "e\u0301e\u0301", "か\u3099き\u3099", "ｶﾞｷﾞ", "㍿㍿ ㍿", "①②③ x", "　　", "\u00a0",
            "x　", "　x", "a\u200bb", b"\xe3\x81", b"a\xffb\xe3\x81\x8b", b"\xe3\x81\x8b\xe3",  "猫" * 40 + "Ａ", "▁a▁", "и\u0306й", "ǅ ǆ"]
    ht, ho = synth.pack([x.encode("utf-8", errors="surrogateescape") if isinstance(x, str) else x for x in hand])
    for classes in ("", None):
        h = emu.load(blob, env=dict(env), **({"classes": classes} if classes is not None else {}))
        for tx, ox in ((mt, mo), (ft, fo), (ht, ho)):
            ids, io = h.encode_batch(tx, ox)
            assert h.status == 0
            oids, oio = o.encode_batch(tx, ox)
            np.testing.assert_array_equal(io, oio)
            np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["test_model", "uni1k_bf", "uni1k_suffix", "uni1k_ident", "uni32k"])
@pytest.mark.parametrize("env", [{}, {"SPMX_NO_COMPRESS": "1"}, {"SPMX_NO_FAST": "1"},
                                 {"SPMX_NO_COMPRESS": "1", "SPMX_NO_FAST": "1"}])
def test_emu_tile_variants(model, env, emu, oracle, corpora, monkeypatch):
    """One-byte space symbol on/off x ASCII fast path on/off: same ids; with both on, ASCII sentences are normalized
    by the tile that drew them and stray non-ASCII ones are set aside on the class's hard list."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    blob = fixtures.model_blob(model)
    h = emu.load(blob)
    assert bool(h.flags() & K_COMPRESS) == ("SPMX_NO_COMPRESS" not in env)
    o = oracle.load(blob)
    for name, k in (("edge", 10 ** 6), ("synth20k", 300), ("mixed2k", 40), ("botchan", 150)):
        text, offs = fixtures.head(*corpora[name], k)
        ids, io = h.encode_batch(text, offs)
        assert h.status == 0
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
        hard = h.path()["backlog"]
        if not env and name == "synth20k":       # ASCII sentences stay with the tile that drew them
            assert hard < 0.1 * (len(offs) - 1)
        if "SPMX_NO_FAST" in env or model == "uni1k_suffix":   # every lane in norm_lane_any: nothing is set aside
            assert hard == 0


@pytest.mark.parametrize("model", ["uni32k", "bpe32k", "uni1k_bf"])
def test_emu_backlog(model, emu, oracle, corpora):
    """Full 64-lane ASCII tiles (one wavefront, so every tile is full): the few sentences that need the general
    normalizer wait in the wave's backlog and run as tiles of their own; same ids."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    # (the general streaming kernel is what is under test: the word kernels, which would take most of these sentences
    # first, are switched off)
    h = emu.load(blob, cus=1, classes=None, env={"SPMX_TILE_WAVES": "1", "SPMX_NO_WORD_KERNEL": "1"})
    o = oracle.load(blob)
    t1, o1 = fixtures.head(*corpora["synth20k"], 1500)
    t2, o2 = corpora["edge"]
    tb1, tb2 = np.asarray(t1).tobytes(), np.asarray(t2).tobytes()
    sents = [tb1[int(o1[i]):int(o1[i + 1])] for i in range(len(o1) - 1)]
    extra = [tb2[int(o2[i]):int(o2[i + 1])] for i in range(len(o2) - 1)]
    for k, e in enumerate(extra):                      # sprinkle the edge cases over the ASCII sentences
        sents.insert(7 + 19 * k, e)
    text, offs = synth.pack(sents)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    assert h.path()["backlog"] >= 3


K_WORDWISE = 1 << 10   # dev.h kNfBpeWordwise


@pytest.mark.parametrize("model,corpus,k", [("test_ja_model", "ja", 200), ("c5_250k", "mixed2k", 200),
                                             ("uni1k_bf", "edge", 10 ** 6), ("bpe32k", "mixed2k", 40)])
@pytest.mark.parametrize("env", [{}, {"SPMX_NO_LANE_GENERAL": "1"}])
def test_emu_lane_general_normalizer(model, corpus, k, env, emu, oracle, corpora, monkeypatch):
    """Non-ASCII text: a tile with enough such sentences normalizes them itself (norm_lane_any); with that switched
    off they all go through the hard list.  Same ids."""
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    blob = fixtures.model_blob(model)
    h = emu.load(blob)
    o = oracle.load(blob)
    text, offs = fixtures.head(*corpora[corpus], k)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    assert h.path()["backlog"] <= len(offs) - 1      # (only full 64-lane tiles use the backlog: few do at this size)


@pytest.mark.parametrize("model", ["bpe1k", "bpe32k", "bpe1k_bf_uds", "bpe1k_noesc", "bpe1k_llama"])
@pytest.mark.parametrize("env", [{}, {"SPMX_NO_WORDWISE": "1"}, {"SPMX_NO_COMPRESS": "1"}, {"SPMX_NO_FAST": "1"},
                                 {"SPMX_NO_STREAM": "1"}, {"SPMX_NO_WAVE": "1"}, {"SPMX_WORDTAB_MIN": "1"},
                                 {"SPMX_NO_WORDTAB": "1"}])
def test_emu_bpe_variants(model, env, emu, oracle, corpora, monkeypatch):
    """BPE: lane-per-sentence word-by-word form (word-wise models) vs sentence-per-wave form: same ids; sentences
    with a word longer than the lane form's slots take the long form."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    blob = fixtures.model_blob(model)
    h = emu.load(blob)
    wordwise = bool(h.flags() & K_WORDWISE)
    assert wordwise == (model in ("bpe1k", "bpe32k", "bpe1k_llama") and "SPMX_NO_WORDWISE" not in env
                        and "SPMX_NO_COMPRESS" not in env)
    o = oracle.load(blob)
    for name, k in (("edge", 10 ** 6), ("synth20k", 200), ("mixed2k", 40), ("botchan", 120)):
        text, offs = fixtures.head(*corpora[name], k)
        ids, io = h.encode_batch(text, offs)
        assert h.status == 0
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
        if name == "edge" and wordwise and "SPMX_NO_STREAM" not in env:
            assert h.path()["long"] >= 3      # "a" * 300, "0123456789" * 40, a long CJK run ...
        if name == "synth20k" and wordwise and not env:
            assert h.path()["backlog"] < 0.1 * (len(offs) - 1) and h.path()["long"] == 0


@pytest.mark.parametrize("model,opts", [("test_model", "bos:eos"), ("test_model", "reverse:bos"),
                                         ("bpe1k", "eos:reverse:bos"), ("uni1k_bf", "reverse")])
def test_emu_extra_options(model, opts, emu, oracle, corpora):
    blob = fixtures.model_blob(model)
    h = emu.load(blob)
    o = oracle.load(blob)
    h.set_encode_extra_options(opts)
    o.set_encode_extra_options(opts)
    text, offs = corpora["edge"]
    ids, io = h.encode_batch(text, offs)
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["uni1k", "bpe1k"])
def test_emu_set_vocabulary(model, emu, oracle, corpora):
    """UNUSED pieces: skipped by the unigram walk, resegmented by BPE
    (unigram_model_test.cc:873-928, bpe_model_test.cc:195-250)."""
    import sentencepiece as spm   # only to list piece strings
    blob = fixtures.model_blob(model)
    sp = spm.SentencePieceProcessor(model_proto=blob)
    vocab = [sp.id_to_piece(i) for i in range(0, sp.get_piece_size(), 3)]
    h = emu.load(blob)
    o = oracle.load(blob)
    text, offs = fixtures.head(*corpora["botchan"], 150)
    h.set_vocabulary(vocab)
    o.set_vocabulary(vocab)
    ids, io = h.encode_batch(text, offs)
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    h.reset_vocabulary()
    o.reset_vocabulary()
    ids, io = h.encode_batch(text, offs)
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(ids, oids)


def test_emu_capacity_report(emu, corpora):
    """Too small an output buffer: nothing is written past it and the needed size comes back."""
    import ctypes as C
    h = emu.load(fixtures.model_blob("test_model"))
    text, offs = fixtures.head(*corpora["botchan"], 50)
    full, io = h.encode_batch(text, offs)
    ids = np.full(8, -7, dtype=np.int32)
    id_offs = np.zeros(len(offs), dtype=np.uint64)
    tot = C.c_uint64(0)
    text, offs = np.ascontiguousarray(text), np.ascontiguousarray(offs)
    rc = h.lib.spmx_encode_batch_device(h.sp._h, text.ctypes.data, len(text), offs.ctypes.data, len(offs) - 1, ids.ctypes.data, 8,
                                        id_offs.ctypes.data, None, C.byref(tot))
    assert rc == 8 and tot.value == len(full)            # RESOURCE_EXHAUSTED + the needed capacity
    assert (ids == -7).all()
    np.testing.assert_array_equal(id_offs, io)


@pytest.mark.parametrize("model", ["test_model", "c5_250k_bf", "test_ja_model"])
def test_emu_document_length(model, emu, oracle, corpora):
    """Document-length sentences: the second streaming launch (classes above 16 KiB), per-lane normalizers for ASCII and
    for everything else."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h = emu.load(blob)
    o = oracle.load(blob)
    bot, boffs = corpora["botchan"]
    ja, joffs = corpora["ja"]
    docs = [bot[:int(boffs[220])].tobytes().replace(b"\n", b" "),          # ~14 KB of ASCII
            ja[:int(joffs[60])].tobytes(),                                  # Japanese, well over 8 KB
            b"x" * 9000, ("猫 " * 3000).encode()]
    assert all(len(d) > 8192 for d in docs)
    text, offs = synth.pack(docs)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)


def long_documents(corpora, size):
    """Documents of about `size` bytes + a few shapes that stress the capacities: a word of thousands of characters,
    NFKC expansions (U+FDFA -> 18 characters), malformed bytes (each becomes U+FFFD), leading whitespace, nothing."""
    from sentencepiece_amd import synth
    bot, _ = corpora["botchan"]
    ja, _ = corpora["ja"]
    docs = [bot[:size].tobytes(), ja[:size * 2 // 3].tobytes(), b"x" * (size // 8) + b" " + b"0123456789" * (size // 40),
            bot[:300].tobytes(), b"", ("\ufdfa" * (size // 30)).encode(), b"\xff\xfe" * (size // 20),
            (" " * (size // 10) + "a").encode()]
    return synth.pack(docs)


@pytest.mark.parametrize("word_norm", ["off", "on"])
@pytest.mark.parametrize("classes", ["small", "default"])
@pytest.mark.parametrize("model", ["uni1k_uds", "uni1k_suffix", "bpe1k_noesc", "bpe1k_bf_uds", "test_model", "bpe1k"])
def test_emu_no_length_limit(model, classes, word_norm, emu, oracle, corpora):
    """Models the fast forms take only in part (user-defined symbols, whitespace as suffix, BPE pieces that span
    words) and documents far beyond every staged class: every sentence is encoded, bit-equal to the oracle.  The
    shrunken class table sends them through the overflow launch (exact capacities), the default one through the
    document launch; BPE takes the long form."""
    from tests import emulib
    blob = fixtures.model_blob(model)
    # (SPMX_NO_WORD_NORM: the word rounds leave text that is not plain ASCII to the general kernels, whose overflow /
    # document launches are what this test is about)
    # word_norm "on": the default path -- the word rounds normalize such words themselves and hand the documents on to the
    # tail's wavefront form
    h = emu.load(blob, classes=emulib.SMALL_CLASSES if classes == "small" else None, env={"SPMX_NO_WORD_NORM": "1"} if word_norm == "off" else None)
    o = oracle.load(blob)
    text, offs = long_documents(corpora, 24000)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0 and not h.sent_status.any()
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    if word_norm == "off" and (model.startswith("uni") or model == "test_model"):
        assert h.path()["overflow"] >= 1          # the NFKC expansions outgrow any class column


@pytest.mark.parametrize("big", ["64", "5000", "0"])
@pytest.mark.parametrize("model", ["test_model", "bpe1k"])
def test_emu_compaction_of_document_blocks(model, big, emu, oracle, corpora):
    """Blocks of 64 sentences with many ids (documents) are compacted by a launch of their own (kernels.h
    compact_big_block): items of 8192 ids dealt to every wave.  A mix of documents and short sentences over several
    blocks, ids and the spans form, with the threshold at 64 ids (nearly every block), 5000 and off."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob, classes=None, env={"SPMX_COMPACT_BIG": big}), oracle.load(blob)
    bot, boffs = corpora["botchan"]
    rng = np.random.default_rng(7)
    docs = []
    for i in range(150):
        a = int(rng.integers(0, len(boffs) - 400))
        k = int(rng.choice([1, 1, 1, 2, 40, 300]))
        docs.append(bot[int(boffs[a]):int(boffs[a + k])].tobytes().replace(b"\n", b" "))
    docs[70] = b""
    text, offs = synth.pack(docs)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    got = h.encode_spans(text, offs)
    want = o.encode_spans(text, offs)
    for a, b, nm in zip(got, want, ("ids", "begin", "end", "id_offsets")):
        np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64), err_msg=nm)


def test_emu_hundred_kilobyte_documents(emu, oracle, corpora):
    """100 KB documents through a user-defined-symbol unigram model and a BPE model whose pieces span words."""
    for model in ("uni1k_uds", "bpe1k_noesc"):
        blob = fixtures.model_blob(model)
        h, o = emu.load(blob, classes=None), oracle.load(blob)
        text, offs = long_documents(corpora, 100_000)
        ids, io = h.encode_batch(text, offs)
        assert not h.sent_status.any()
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["test_model", "uni1k_uds", "uni1k_suffix", "bpe1k", "bpe1k_noesc"])
def test_emu_spans_and_normalize_of_long_documents(model, emu, oracle, corpora):
    """The spans form and Normalize(input, &normalized, &norm_to_orig) beyond the staged classes: the lane-per-sentence
    align / normalize kernels (kernels_long.h), which store nothing per sentence."""
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob, classes=None), oracle.load(blob)
    text, offs = long_documents(corpora, 20000)
    for opts in ("", "reverse:bos:eos"):
        h.set_encode_extra_options(opts)
        o.set_encode_extra_options(opts)
        got = h.encode_spans(text, offs)
        want = o.encode_spans(text, offs)
        for a, b, nm in zip(got, want, ("ids", "begin", "end", "id_offsets")):
            np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64), err_msg="%s [%s]" % (nm, opts))
    h.set_encode_extra_options("")
    o.set_encode_extra_options("")
    gn = h.normalize_batch(text, offs)
    wn = o.normalize_batch(text, offs)
    for a, b, nm in zip(gn, wn, ("normalized", "norm_offsets", "norm_to_orig")):
        np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64), err_msg=nm)


def test_emu_bpe_long_form_with_unused_pieces(emu, oracle, corpora):
    """SetVocabulary turns most pieces UNUSED: every sentence resegments through rev_merge (bpe_model.cc:175-200); the
    sentence-per-wave form hands what outgrows its 64-entry table to the long form, which has no such bound."""
    import sentencepiece as spm   # only to list piece strings
    blob = fixtures.model_blob("bpe1k")
    sp = spm.SentencePieceProcessor(model_proto=blob)
    vocab = [sp.id_to_piece(i) for i in range(0, sp.get_piece_size(), 7)]
    bot, boffs = corpora["botchan"]
    from sentencepiece_amd import synth
    text, offs = synth.pack([bot[:3000].tobytes(), bot[3000:3500].tobytes(), bot[:12000].tobytes()])
    for env in ({}, {"SPMX_NO_WAVE": "1"}):
        h, o = emu.load(blob, env=env), oracle.load(blob)
        h.set_vocabulary(vocab)
        o.set_vocabulary(vocab)
        ids, io = h.encode_batch(text, offs)
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
        assert not h.sent_status.any()


def test_emu_per_sentence_status(emu, oracle, corpora):
    """A sentence the reference's Encode fails -- a character that is a CONTROL piece among the BPE symbols consumes no
    text, so the reference reports "all normalized characters are not consumed" (sentencepiece_processor.cc:566-571,
    :628) -- yields no ids and a status byte; the other sentences of the batch are encoded as ever, and the
    single-sentence call returns that Status."""
    import ctypes as C
    from sentencepiece import sentencepiece_model_pb2 as pb
    from sentencepiece_amd import synth
    m = pb.ModelProto()
    m.ParseFromString(fixtures.model_blob("bpe1k"))
    p = m.pieces.add()
    p.piece, p.score, p.type = "\u2603", 0.0, 3            # a one-character CONTROL piece
    blob = m.SerializeToString()
    h, o = emu.load(blob), oracle.load(fixtures.model_blob("bpe1k"))
    bot, boffs = corpora["botchan"]
    good = [bot[int(boffs[i]):int(boffs[i + 1])].tobytes() for i in range(5)]
    bad = "say \u2603 then".encode()
    text, offs = synth.pack(good[:2] + [bad] + good[2:])
    ids, io = h.encode_batch(text, offs)
    oids, oio = o.encode_batch(*synth.pack(good))
    assert h.sent_status.tolist() == [0, 0, 13, 0, 0, 0] and h.status == 1
    cnt = np.diff(io.astype(np.int64))
    assert cnt[2] == 0
    np.testing.assert_array_equal(np.delete(cnt, 2), np.diff(oio.astype(np.int64)))
    np.testing.assert_array_equal(ids, oids)
    out = np.zeros(64, dtype=np.int32)
    n_ids = C.c_uint64(0)
    assert h.lib.spmx_encode(h.sp._h, bad, len(bad), out.ctypes.data, 64, C.byref(n_ids)) == 13
    assert "not consumed" in h.lib.spmx_last_error(None).decode()
    from tests import refshim
    if refshim.available():                                # the compiled reference fails the same sentence
        r = refshim.RefLib().load(blob)
        with pytest.raises(RuntimeError):
            r.encode(bad)
        assert r.encode(good[0]).tolist() == ids[:int(io[1])].tolist()


@pytest.mark.parametrize("model", ["uni32k", "uni1k_ident", "bpe1k", "uni1k_bf"])
def test_emu_fast_keeps_identity_characters(model, emu, oracle):
    """ASCII sentences with a stray character that starts no charsmap key (e-acute, CJK, emoji ...), a malformed
    byte or a truncated character stay in the FAST kernel's own normalizer; compatibility characters, U+3000 and a
    literal U+2581 go to the general normalizers.  Same ids either way."""
    from sentencepiece_amd import synth
    rng = np.random.default_rng(5)
    words = [b"hello", b"world", b"the", b"cat", b"sat", b"on", b"a", b"mat"]
    keep = ["é", "ü", "日本", "€", "\U0001f600", "ñ"]
    leave = ["ＡＢ", "　", "▁", "㍿"]
    sent, n_leave = [], 0
    for i in range(300):
        ws = [words[int(k)] for k in rng.integers(0, len(words), size=int(rng.integers(1, 30)))]
        r = rng.random()
        if r < 0.4:
            ws.insert(int(rng.integers(0, len(ws) + 1)), keep[int(rng.integers(0, len(keep)))].encode())
        elif r < 0.5:
            ws.insert(int(rng.integers(0, len(ws) + 1)), leave[int(rng.integers(0, len(leave)))].encode())
            n_leave += 1
        elif r < 0.7:
            ws.insert(int(rng.integers(0, len(ws) + 1)), bytes([int(rng.integers(0x80, 0x100))]))
        elif r < 0.8:
            ws.append("日".encode()[:int(rng.integers(1, 3))])      # truncated at the end of the sentence
        sent.append(b" ".join(ws))
    text, offs = synth.pack(sent)
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob), oracle.load(blob)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    assert h.path()["backlog"] <= 0.3 * 300     # (set aside only when the tile has too few such sentences to keep them)


@pytest.mark.parametrize("model", ["bpe1k", "bpe1k_llama"])
def test_emu_bpe_documents(model, emu, oracle):
    """BPE documents: the lane form word by word; a document with a word that outgrows the LDS slots takes the long form."""
    from tests.test_gpu_parity import bpe_documents
    text, offs = bpe_documents()
    text, offs = fixtures.head(text, offs, 8)
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob), oracle.load(blob)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["test_model", "bpe1k"])
def test_emu_first_offset_not_rebased(model, emu, oracle, corpora):
    """offsets[0] != 0 (and not a multiple of 16): the host forms stage text[offsets[0]:] and the kernels address
    text + offsets[i] as given."""
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob), oracle.load(blob)
    text, offs = fixtures.head(*corpora["botchan"], 200)
    want, wio = o.encode_batch(text, offs)
    for shift in (1, 7, 16, 33):
        pad = np.concatenate([np.full(shift, 0xE3, dtype=np.uint8), text])
        ids, io = h.encode_batch(pad, offs + np.uint64(shift))
        np.testing.assert_array_equal(io, wio)
        np.testing.assert_array_equal(ids, want)


@pytest.mark.parametrize("model", ["test_model", "uni1k_uds", "bpe1k", "bpe1k_noesc", "bpe1k_bf_uds"])
def test_emu_arena_overflow_and_retry(model, emu, oracle, corpora):
    """The id arena of a call is sized by an estimate; a batch that outgrows it is encoded again with the arena the first
    attempt asked for.  SPMX_ARENA_FIRST forces that path for every kernel family (stream lanes, sentence per wave, long
    form): same ids, and -- the round-2 bug -- nothing compacted from beyond the arena's end in between (the status comes
    back before the compaction is launched; ASAN on the emulator and a memory fault on the GPU found it)."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h = emu.load(blob, env={"SPMX_ARENA_FIRST": "600"})
    o = oracle.load(blob)
    bot, boffs = corpora["botchan"]
    docs = [bot[:int(boffs[400])].tobytes().replace(b"\n", b" "), b"ab " * 3000, b"x" * 2500 + b" 0123456789", b"", b"short one"]
    text, offs = synth.pack(docs)
    ids, io = h.encode_batch(text, offs)
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    text, offs = fixtures.head(*corpora["botchan"], 300)        # many short sentences: the staged classes
    ids, io = h.encode_batch(text, offs)
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    got = h.encode_spans(text, offs)
    want = o.encode_spans(text, offs)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64))


@pytest.mark.parametrize("model", ["test_model", "test_ja_model", "uni1k", "uni1k_bf", "uni1k_uds", "uni1k_ident", "uni1k_suffix",
                                   "uni32k", "c5_250k_bf"])
@pytest.mark.parametrize("fold", ["float", "exact"])
def test_emu_wave_cooperative_unigram(model, fold, emu, oracle, corpora):
    """kernels_uniwave.h: a sentence per wavefront -- parallel trie walks from 64 character starts, the relaxations
    folded with the scores in registers, in the reference's order.  Every unigram model, every corpus, with the word
    kernels off so that all sentences come here (SPMX_UNI_WAVE_MAX: classes below that many sentences take this form);
    the fold in float arithmetic where the model allows it (uw_fold_float; ties re-folded exactly) and in the
    reference's double arithmetic throughout (uw_fold_exact)."""
    blob = fixtures.model_blob(model)
    env = {"SPMX_UNI_WAVE_MAX": "1000000", "SPMX_NO_WORD_KERNEL": "1"}
    if fold == "exact":
        env["SPMX_UW_EXACT"] = "1"
    h = emu.load(blob, env=env)
    o = oracle.load(blob)
    for name, k in (("edge", 10 ** 6), ("botchan", 150), ("mixed2k", 40), ("ja", 40), ("synth20k", 150)):
        text, offs = fixtures.head(*corpora[name], k)
        for opts in ("", "bos:eos:reverse"):
            h.sp.SetEncodeExtraOptions(opts)
            o.set_encode_extra_options(opts)
            ids, io = h.encode_batch(text, offs)
            assert h.status == 0 and not h.sent_status.any()
            oids, oio = o.encode_batch(text, offs)
            np.testing.assert_array_equal(io, oio)
            np.testing.assert_array_equal(ids, oids)
        h.sp.SetEncodeExtraOptions("")
        o.set_encode_extra_options("")
    assert any(c["kernel"] == "UniLongKernel" for c in h.sp.LastProfile()["classes"])


@pytest.mark.parametrize("quantum", [1.0, 0.25])
@pytest.mark.parametrize("model", ["test_model", "uni32k"])
def test_emu_wave_cooperative_unigram_ties(model, quantum, emu, oracle, corpora):
    """The float fold's tie rule: with every score a multiple of 1 or 1/4, different paths reach a position with EQUAL
    rounded sums all the time -- the reference keeps the first (strict >), and so must the chunk that is folded again
    in double arithmetic after the float fold recorded a tie."""
    from sentencepiece_amd import synth
    blob = synth.requantized_model(fixtures.model_blob(model), quantum)
    h = emu.load(blob, env={"SPMX_UNI_WAVE_MAX": "1000000", "SPMX_NO_WORD_KERNEL": "1"})
    o = oracle.load(blob)
    for name, k in (("botchan", 200), ("mixed2k", 30), ("synth20k", 100)):
        text, offs = fixtures.head(*corpora[name], k)
        ids, io = h.encode_batch(text, offs)
        assert h.status == 0 and not h.sent_status.any()
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("fold", ["float", "exact"])
def test_emu_wave_cooperative_unigram_large_scores(fold, emu, oracle, corpora):
    """Deep inside a megabyte document best_path_score is 10^6 and a float holds 1/8: the rounded sums of different
    paths tie all the time and the reference's double comparison decides by what the rounding dropped.  The same
    regime in 30 KB: a model whose scores are 64 times test_model's (the float fold's tie-deciding flavour,
    uw_relax_float<true>, and the credit that switches to it)."""
    from sentencepiece_amd import synth
    blob = synth.rescored_model(fixtures.model_blob("test_model"), lambda v: v * 64.0 + 0.001)
    env = {"SPMX_UW_EXACT": "1"} if fold == "exact" else None
    h, o = emu.load(blob, classes=None, env=env), oracle.load(blob)
    bot, boffs = corpora["botchan"]
    docs = [bot[int(boffs[a]):int(boffs[a + 420])].tobytes().replace(b"\n", b" ") for a in (0, 500, 1000)] + ["猫 も 杓子 も ".encode() * 900]
    text, offs = synth.pack(docs)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0 and not h.sent_status.any()
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["uni32k", "test_model"])
def test_emu_word_kernels_then_wave_form(model, emu, oracle, corpora):
    """The word kernels first, what they leave through the wave-cooperative form: the default order on the device."""
    blob = fixtures.model_blob(model)
    h = emu.load(blob, classes=None, env={"SPMX_UNI_WAVE_MAX": "1000000"})
    o = oracle.load(blob)
    for name, k in (("synth20k", 1500), ("edge", 10 ** 6), ("botchan", 300)):
        text, offs = fixtures.head(*corpora[name], k)
        ids, io = h.encode_batch(text, offs)
        assert h.status == 0
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["uni32k_w16", "bpe32k"])
@pytest.mark.parametrize("env", [{"SPMX_DYN_SLOTS_LOG2": "5", "SPMX_DYN_LIST_CAP": "7"},
                                 {"SPMX_DYN_SLOTS_LOG2": "12", "SPMX_DYN_LIST_CAP": "40"}])
def test_emu_call_local_word_memo_overflows(model, env, emu, oracle):
    """The call-local word memo (kernels_word.h) with room for a handful of words: a table whose probes run out and a
    word list that is full leave their sentences to the later rounds / the general kernels -- same ids.  (A corpus of
    real text has more distinct words than the default 2^18 a call may enter.)"""
    import bench
    blob = bench.model_blob(model)
    text, offs = bench.corpus_for(model if model.endswith("_w16") else "uni32k", 5000, 20250227, False)
    e = dict(env)
    e["SPMX_FORCE_WORD_DP"] = "0"
    h = emu.load(blob, classes="", cus=4, env=e)
    h.sp.SetProfiling(True)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0
    kernels = {c["kernel"]: c["sentences"] for c in h.sp.LastProfile()["classes"] if c["kernel"]}
    assert any(k.startswith(("EncodeWordCollect", "EncodeWordWaveCollect")) for k in kernels)   # the word rounds ran ...
    assert sum(v for k, v in kernels.items() if "Word" not in k) > 0   # ... and left work to the other kernels
    oids, oio = oracle.load(blob).encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("waves", ["16", "5", "1"])
def test_emu_word_kernels_at_other_workgroup_widths(waves, emu, oracle):
    """The word kernels' wavefronts per workgroup (SPMX_WORD_WAVES; 12 by default, measured) only shape the launch."""
    import bench
    blob = bench.model_blob("uni32k")
    text, offs = bench.corpus_for("uni32k", 4500, 20250301, False)
    h = emu.load(blob, classes="", cus=3, env={"SPMX_WORD_WAVES": waves, "SPMX_FORCE_WORD_DP": "0"})
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0
    oids, oio = oracle.load(blob).encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["uni32k", "bpe32k", "bpe1k_llama", "uni1k_uds"])
def test_offsets_beyond_four_gigabytes(model, oracle):
    """The device form with offsets that do not start at 0 and do not fit 32 bits (a slice of a corpus of more than 4 GB
    resident in HBM: the text pointer is the slice's address minus offsets[0], as the host forms pass it): every kernel
    on the way -- plain scan, classify, word rounds, general launch, scan, compact -- does its offset arithmetic in 64 bits."""
    import ctypes as C
    from sentencepiece_amd import synth
    from tests import emulib, wordfuzz
    em = emulib.EmuLib()
    lib = em.lib
    blob = fixtures.model_blob(model)
    h = em.load(blob, classes=None)
    text, offs = synth.ascii_corpus(3000, seed=9)
    oi, oo = oracle.load(blob).encode_batch(text, offs)
    n = len(offs) - 1
    lib.spmx_encode_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                             C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    for base in ((1 << 32) - 3, (1 << 33) + 7, (1 << 40) + 1):
        buf = np.zeros(len(text) + 64, dtype=np.uint8)
        buf[16:16 + len(text)] = text
        o2 = offs.astype(np.uint64) + np.uint64(base)
        ptr = (buf.ctypes.data + 16 - base) & ((1 << 64) - 1)
        ids = np.zeros(len(text) + 64, dtype=np.int32)
        io = np.zeros(n + 1, dtype=np.uint64)
        tot = C.c_uint64(0)
        rc = lib.spmx_encode_batch_device(h.sp._h, C.c_void_p(ptr), len(text), o2.ctypes.data, n, ids.ctypes.data, len(ids),
                                          io.ctypes.data, None, C.byref(tot))
        assert rc == 0, (base, lib.spmx_last_error(None))
        assert wordfuzz.first_difference(ids[:tot.value], io, np.asarray(oi), np.asarray(oo)) < 0, base


@pytest.mark.parametrize("model,shift", [("uni1k", 0), ("uni1k", 1), ("bpe1k", 0)])
def test_emu_id_offsets_over_whole_scan_tiles(model, shift, emu, oracle, corpora):
    """More than two whole tiles of the id-count scan (kernels.h kScanTile = 2048 sentences): whole tiles take the
    coalesced row form (two counts a lane, offsets stored 16 bytes a lane), the last tile and outputs that are not
    16-byte aligned (shift 1: the caller's offsets array starts 8 bytes off) the lane-owns-32 form."""
    import torch
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob), oracle.load(blob)
    text, offs = corpora["botchan"]
    assert len(offs) - 1 > 2 * 2048
    want, wio = o.encode_batch(text, offs)
    n = len(offs) - 1
    tb = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text)
    pad = np.zeros(len(tb) + 48, dtype=np.uint8)        # (whole 16-byte units around the text, as a device allocation has: spmx.h)
    t0 = 16 + (-pad.ctypes.data) % 16
    pad[t0:t0 + len(tb)] = tb
    d_text = torch.from_numpy(pad)[t0:t0 + len(tb)]
    d_offs = torch.from_numpy(np.ascontiguousarray(offs).view(np.int64))
    raw = torch.zeros(n + 1 + 3, dtype=torch.int64)
    base = (-(raw.data_ptr() // 8)) % 2          # first element that is 16-byte aligned
    io = raw[base + shift: base + shift + n + 1]
    assert (io.data_ptr() % 16 == 0) == (shift == 0)
    ids = torch.empty(len(want) + 64, dtype=torch.int32)
    _, io2, tot = h.sp.EncodeDevice(d_text, d_offs, ids, io)
    assert int(tot) == len(want)
    np.testing.assert_array_equal(io.numpy().view(np.uint64), wio)
    np.testing.assert_array_equal(ids[:len(want)].numpy(), want)


@pytest.mark.parametrize("model", ["uni32k", "bpe32k"])
@pytest.mark.parametrize("cap", ["0", "256", "512", "1024", "4096"])
def test_emu_compact_image_sizes(model, cap, emu, oracle, corpora, monkeypatch):
    """CompactKernel's LDS image at other sizes (kernels.h compact_block): a block of 64 sentences moves as one part, two
    halves, four quarters -- the fewest whose ids fit -- or by the search form (0, and blocks no quarter of which fits);
    blocks with a sentence of 32-bit ids (what the tail kernels wrote) take the search form too."""
    monkeypatch.setenv("SPMX_COMPACT_STAGED", cap)
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob), oracle.load(blob)
    for name, k in (("botchan", 330), ("edge", 10 ** 6), ("mixed2k", 70)):
        text, offs = fixtures.head(*corpora[name], k)
        ids, io = h.encode_batch(text, offs)
        assert h.status == 0
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


SPLIT_MODELS = ["test_model", "test_ja_model", "uni1k", "uni1k_bf", "uni1k_ident", "uni1k_suffix", "uni32k", "uni32k_w16",
                "c5_250k", "c5_250k_bf"]


@pytest.mark.parametrize("model", SPLIT_MODELS)
def test_emu_split_form(model, emu, oracle, corpora):
    """kernels_matchfold.h: EncodeOptimized's trie walks (wave-cooperative, a start per lane) and its fold (a sentence
    per lane, from the candidate stream) taken apart.  SPMX_SPLIT_MIN=0 sends every class of up to 4096 raw bytes there;
    the word kernels are off so that all sentences come.  Every corpus, with and without extra options, the spans form,
    and the longest mixed-script sentences (the classes the form exists for)."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h = emu.load(blob, env={"SPMX_SPLIT_MIN": "0", "SPMX_NO_WORD_KERNEL": "1"})
    o = oracle.load(blob)
    t, of = corpora["mixed2k"]
    n = len(of) - 1
    pick = np.concatenate([np.arange(n - 4, n), np.arange(n - 400, n - 5, 100)])
    batches = [fixtures.head(*corpora[name], k) for name, k in (("edge", 10 ** 6), ("botchan", 150), ("mixed2k", 40), ("ja", 20), ("synth20k", 100))]
    batches.append(synth.gather_packed(t, of, pick))
    for text, offs in batches:
        for opts in ("", "bos:eos:reverse"):
            h.sp.SetEncodeExtraOptions(opts)
            o.set_encode_extra_options(opts)
            ids, io = h.encode_batch(text, offs)
            assert h.status == 0
            oids, oio = o.encode_batch(text, offs)
            np.testing.assert_array_equal(io, oio)
            np.testing.assert_array_equal(ids, oids)
        h.sp.SetEncodeExtraOptions("")
        o.set_encode_extra_options("")
    text, offs = fixtures.head(*corpora["mixed2k"], 60)
    got = h.encode_spans(text, offs)
    want = o.encode_spans(text, offs)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64))


@pytest.mark.parametrize("model", ["uni32k", "c5_250k_bf"])
def test_emu_split_form_stream_overflow(model, emu, oracle, corpora):
    """A sentence whose candidate stream outgrows its share of the slab (SPMX_SPLIT_CANDS=1: one candidate per normalized
    byte) leaves the split form for the call's overflow launch -- the lane-per-sentence kernel -- with the same ids; so
    does one whose normalized text outgrows its column (NFKC expansions)."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h = emu.load(blob, env={"SPMX_SPLIT_MIN": "0", "SPMX_NO_WORD_KERNEL": "1", "SPMX_SPLIT_CANDS": "1"})
    o = oracle.load(blob)
    docs = [b"the thing that there is " * 40, "㍿㌀㌁ ".encode() * 60, b"a", b"", "日本語のテキスト ".encode() * 30]
    text, offs = synth.pack(docs)
    t2, o2 = fixtures.head(*corpora["synth20k"], 80)
    over = 0
    for tx, ox in ((text, offs), (t2, o2)):
        ids, io = h.encode_batch(tx, ox)
        assert h.status == 0
        over += h.path()["overflow"]
        oids, oio = o.encode_batch(tx, ox)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
    assert over > 0


def test_emu_split_form_is_the_default_for_long_classes(emu, oracle, corpora):
    """Without any switch the classes beyond 576 raw bytes of an eligible unigram model take the split form: the longest
    sentences of the mixed-script corpus through the default plan (class table and thresholds as shipped)."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob("c5_250k")
    o = oracle.load(blob)
    t, of = corpora["mixed2k"]
    n = len(of) - 1
    text, offs = synth.gather_packed(t, of, np.arange(n - 24, n))
    oids, oio = o.encode_batch(text, offs)
    for own in ("0", "1"):          # split tiles beside lane tiles in one launch (the default), or in a launch of their own
        h = emu.load(blob, classes=None, env={"SPMX_NO_WORD_KERNEL": "1", "SPMX_SPLIT_LAUNCH": own})
        ids, io = h.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
        prof = h.sp.LastProfile()["classes"]
        if own == "1":
            split = [c for c in prof if c["kernel"].startswith("EncodeSplit")]
            assert split and split[0]["sentences"] == 24, prof
        else:
            # the fold counts blocks of 16 candidates, the lane-per-sentence search one probe per iteration
            trips = [c["phase_cycles"]["search_trips"] for c in prof if c["kernel"].startswith("EncodeStream")]
            assert trips and max(trips) < 4096 * 2, trips
