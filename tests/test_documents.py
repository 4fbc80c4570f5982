"""Documents through the wave-cooperative form (csrc/kernels_uniwave.h) and the compaction of their blocks (kernels.h
compact_big_block), on the CPU model of the wavefront and -- the -m gpu twin of every test -- on the device: the cases round 6's
register fold added to the ones tests/test_emu.py and tests/test_gpu_parity.py hold.  Reference: unigram::Model::EncodeOptimized
(src/unigram_model.cc:889-1020); the checker is the oracle (pinned to the compiled reference, tests/test_oracle.py)."""
import numpy as np
import pytest

from tests import emulib, fixtures


@pytest.fixture(scope="module", params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def eng(request):
    return request.param, emulib.backend(request.param)


# the wave-cooperative kernel's two forms (api.cc uw_pipe): a wavefront per document, and a workgroup of two -- a walker
# and a folder -- per document (the default for few long documents)
FORMS = [{"SPMX_UW_PIPE": "0"}, {"SPMX_UW_PIPE": "2"}]


def botchan_docs(corpora, lines_per_doc, n_docs, step=37):
    bot, boffs = corpora["botchan"]
    last = len(boffs) - 1 - lines_per_doc
    return [bot[int(boffs[(i * step) % last]):int(boffs[(i * step) % last + lines_per_doc])].tobytes().replace(b"\n", b" ")
            for i in range(n_docs)]


def check(h, o, docs):
    from sentencepiece_amd import synth
    text, offs = synth.pack(docs)
    ids, io = h.encode_batch(text, offs)
    assert h.status == 0 and not h.sent_status.any()
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    return h


@pytest.mark.parametrize("form", FORMS, ids=["wave", "pair"])
@pytest.mark.parametrize("rescore", ["x1", "q1", "q025", "x64"])
def test_document_fold_tie_regimes(rescore, form, eng, oracle, corpora):
    """The float fold's tie rule (uw_relax_float): scores quantized to 1 and 1/4 (equal sums of different paths all the
    time), and scores 64 times as large (the regime deep inside a megabyte document: the float's granularity is coarser
    than the scores' differences, the reference's double comparison decides by what the rounding dropped; never for the
    UNK candidate -- the documents hold characters the model does not know)."""
    from sentencepiece_amd import synth
    which, lib = eng
    blob = fixtures.model_blob("test_model")
    if rescore == "q1":
        blob = synth.requantized_model(blob, 1.0)
    elif rescore == "q025":
        blob = synth.requantized_model(blob, 0.25)
    elif rescore == "x64":
        blob = synth.rescored_model(blob, lambda v: v * 64.0 + 0.001)
    h, o = lib.load(blob, env=form), oracle.load(blob)
    n = 24 if which == "gpu" else 3
    docs = botchan_docs(corpora, 420 if which == "gpu" else 260, n) + ["猫 も 杓子 も Zürich ".encode() * 700]
    h = check(h, o, docs)
    want = "UniLongPipeKernel" if form["SPMX_UW_PIPE"] == "2" else "UniLongKernel"
    assert any(c["kernel"] == want for c in h.sp.LastProfile()["classes"] if c["kernel"])


@pytest.mark.parametrize("form", FORMS, ids=["wave", "pair"])
@pytest.mark.parametrize("model", ["uni32k_w16", "c5_250k_bf", "test_ja_model"])
def test_document_rows_of_16_and_32_entries(model, form, eng, oracle, corpora):
    """The matrix rows (UniWaveRow): a model whose longest piece has exactly 16 bytes (a piece that fills its row and ends
    one past the chunk), and models with rows of 32; documents cut at every offset of a chunk."""
    import bench
    which, lib = eng
    blob = bench.model_blob(model)
    if model == "uni32k_w16":
        text, offs = bench.corpus_for(model, 4000 if which == "gpu" else 700, 20250227, False)
        raw = text.tobytes()
        cut = [int(offs[i]) for i in range(0, len(offs) - 1, 100)] + [int(offs[-1])]
        docs = [raw[a:b].replace(b"\n", b" ") for a, b in zip(cut[:-1], cut[1:])]
        docs += [docs[0][k:] for k in range(1, 66, 5)]                  # the same text at every phase of the 64-byte chunks
    else:
        ja, joffs = corpora["ja"]
        mixed, moffs = corpora["mixed2k"]
        docs = [ja[:int(joffs[60])].tobytes(), mixed[:int(moffs[30])].tobytes().replace(b"\n", b" ")]
        docs += [docs[0][3 * k:] for k in range(1, 12)]
    assert max(len(d) for d in docs) > 8192
    check(lib.load(blob, env=form), oracle.load(blob), docs)


def test_compaction_of_document_blocks_default_threshold(eng, oracle, corpora):
    """Blocks of 64 sentences with more than 32768 ids go to CompactBigKernel: documents among short sentences, in the
    ids form and in the spans form."""
    from sentencepiece_amd import synth
    which, lib = eng
    blob = fixtures.model_blob("test_model")
    h, o = lib.load(blob), oracle.load(blob)
    bot, boffs = corpora["botchan"]
    rng = np.random.default_rng(11)
    docs = []
    for i in range(192 if which == "gpu" else 130):
        a = int(rng.integers(0, len(boffs) - 700))
        k = 600 if 64 <= i < 128 else int(rng.choice([1, 1, 2, 30]))   # the second block is one of documents
        docs.append(bot[int(boffs[a]):int(boffs[a + k])].tobytes().replace(b"\n", b" "))
    docs[70] = b""
    docs[100] = b"   "                                                  # (nothing but whitespace: the walker's own way out)
    check(h, o, docs)
    text, offs = synth.pack(docs)
    got = h.encode_spans(text, offs)
    want = o.encode_spans(text, offs)
    for a, b, nm in zip(got, want, ("ids", "begin", "end", "id_offsets")):
        np.testing.assert_array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64), err_msg=nm)


def _custom_rule_model(model_type, tmp_path_factory):
    """A throw-away model the pip wheel trains on the spot (test tooling only) with a normalization_rule_tsv of its own: targets
    that begin / end with spaces, that ARE a space, that map to U+2581, that delete, on top of whitespace handling."""
    import io
    import sentencepiece as spm
    d = tmp_path_factory.mktemp("rules")
    tsv = d / "rules.tsv"
    rules = ["41\t20 78 20",          # A -> " x "
             "42\t2581",              # B -> U+2581
             "43 44\t20",             # CD -> " "
             "45\t",                  # E -> (nothing)
             "46\t66 20",             # F -> "f "
             "47\t20 67",             # G -> " g"
             "E9\t65 301",            # e-acute -> e + combining acute
             "3042\t61 20 61"]        # HIRAGANA A -> "a a"
    tsv.write_text("\n".join(rules) + "\n")
    bot = open(fixtures.os.path.join(fixtures.GOLDEN, "botchan.txt"), "rb").read().decode("utf-8", "replace").split("\n")[:3000]
    buf = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(bot), model_writer=buf, vocab_size=600, model_type=model_type,
                                   normalization_rule_tsv=str(tsv), hard_vocab_limit=False)
    return buf.getvalue()


@pytest.mark.parametrize("model_type", ["unigram", "bpe"])
def test_custom_normalization_rules_through_the_default_path(model_type, eng, oracle, corpora, tmp_path_factory):
    """A model with normalization rules of its own (replacements with leading / trailing spaces, a replacement that is the
    space symbol itself, deletions) through the DEFAULT path -- word-local normalization in the word rounds, direct calls --
    as sentences and as documents (ADVICE of round 5: kNfWordLocalNorm assumes a word normalizes independently of its
    neighbours; the loader's check decides whether such a model may take that path at all)."""
    try:
        blob = _custom_rule_model(model_type, tmp_path_factory)
    except Exception as e:
        pytest.skip("the pip wheel cannot train here: %r" % (e,))
    which, lib = eng
    h, o = lib.load(blob), oracle.load(blob)
    rng = np.random.default_rng(5)
    bot, boffs = corpora["botchan"]
    words = [b"A", b"B", b"CD", b"E", b"F", b"G", "é".encode(), "あ".encode(), b"AB", b"xAy", b"BA ", b" F", b"GG", b"ECDE",
             b"C D", b"AF", b"FA", b"  ", b"the", b"CDCD", b"A A", b"B B"]
    docs = []
    for i in range(300 if which == "gpu" else 120):
        a = int(rng.integers(0, len(boffs) - 3))
        base = bot[int(boffs[a]):int(boffs[a + 1])].tobytes().rstrip(b"\r\n").split(b" ")
        out = []
        for w in base:
            out.append(w)
            if rng.random() < 0.35:
                out.append(words[int(rng.integers(0, len(words)))] + (w[:2] if rng.random() < 0.5 else b""))
        docs.append(b" ".join(out))
    docs += [b" ".join(docs[:60]) * 3, b"A", b"B", b"E", b"CD", b" A ", b"EEEE", b""]      # a document; rules alone
    check(h, o, docs)
