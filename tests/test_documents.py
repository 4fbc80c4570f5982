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
