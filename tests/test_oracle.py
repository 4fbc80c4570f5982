"""The oracle (plain-C restatement, oracle/spm_oracle.c) pinned against
 (a) the golden ids the compiled reference produced (tests/golden/), on every
     model x corpus pair of the manifest, and
 (b) the compiled reference itself where oracle/_ref/libspm_ref.so exists
     (this container), including normalizer output."""
import numpy as np
import pytest

from tests import fixtures, refshim


def _keys():
    import json
    import os
    with open(os.path.join(fixtures.GOLDEN, "manifest.json")) as f:
        return sorted(k for k in json.load(f) if not k.startswith("_"))


@pytest.mark.parametrize("key", _keys())
def test_oracle_matches_golden(key, manifest, golden_arrays, corpora, oracle):
    m = manifest[key]
    o = oracle.load(fixtures.model_blob(m["model"]))
    if m["options"]:
        o.set_encode_extra_options(m["options"])
    text, offs = corpora[m["corpus"]]
    ids, io = o.encode_batch(text, offs)
    assert len(ids) == m["tokens"]
    np.testing.assert_array_equal(np.diff(io.astype(np.int64)), golden_arrays[key + "__cnt"].astype(np.int64))
    if key + "__ids" in golden_arrays:
        np.testing.assert_array_equal(ids, golden_arrays[key + "__ids"])
    assert fixtures.sha(ids) == m["sha256"]


def test_botchan_survey_kat(corpora, oracle):
    """95,515 tokens on botchan with the bundled 1k model, and
    the four double/float tie lines come out in the reference's order."""
    o = oracle.load(fixtures.model_blob("test_model"))
    text, offs = corpora["botchan"]
    ids, io = o.encode_batch(text, offs)
    assert len(ids) == 95515
    lines = fixtures.mf.synth.unpack(text, offs)
    for ln in (1754, 1973, 2567, 2998):
        assert b"......." in lines[ln] or b"......" in lines[ln]


@pytest.mark.skipif(not refshim.available(), reason="compiled reference not built (needs /root/reference)")
@pytest.mark.parametrize("model", ["test_model", "test_ja_model", "uni1k_uds", "uni1k_ident", "uni1k_suffix",
                                   "bpe1k_noesc", "bpe1k_bf_uds"])
def test_oracle_normalizer_matches_reference(model, corpora, oracle):
    ref = refshim.RefLib().load(fixtures.model_blob(model))
    o = oracle.load(fixtures.model_blob(model))
    sents = fixtures.mf.edge_sentences() + fixtures.mf.synth.unpack(*fixtures.head(*corpora["mixed2k"], 100))
    for s in sents:
        if b"\x00" in s:
            continue   # the shim passes C strings through ctypes.c_char_p
        assert o.normalize(s) == ref.normalize(s), s


@pytest.mark.skipif(not refshim.available(), reason="compiled reference not built (needs /root/reference)")
def test_oracle_set_vocabulary_matches_reference(corpora, oracle):
    """SetVocabulary / ResetVocabulary (sentencepiece_processor_test.cc:1357-1372 analogue)."""
    for model in ("uni1k", "bpe1k"):
        blob = fixtures.model_blob(model)
        ref = refshim.RefLib().load(blob)
        o = oracle.load(blob)
        text, offs = fixtures.head(*corpora["botchan"], 400)
        import sentencepiece as spm   # only to list the piece strings
        sp = spm.SentencePieceProcessor(model_proto=blob)
        vocab = [sp.id_to_piece(i) for i in range(0, sp.get_piece_size(), 3)]
        ref.set_vocabulary(vocab)
        o.set_vocabulary(vocab)
        a, ao = ref.encode_batch(text, offs)
        b, bo = o.encode_batch(text, offs)
        np.testing.assert_array_equal(ao, bo)
        np.testing.assert_array_equal(a, b)
        ref.reset_vocabulary()
        o.reset_vocabulary()
        a, _ = ref.encode_batch(text, offs)
        b, _ = o.encode_batch(text, offs)
        np.testing.assert_array_equal(a, b)


def test_whole_corpus_checker_counts_differing_sentences(oracle, corpora):
    """tests/fullcheck.py (the all-sentences comparison of the full-size GPU tests and of bench.py's probe): equal CSRs
    pass; an altered id and an altered length are counted, per sentence, and the first one is named."""
    from tests import fullcheck
    blob = fixtures.model_blob("uni32k")
    text, offs = fixtures.head(*corpora["synth20k"], 5000)
    ids, io = oracle.load(blob).encode_batch(text, offs)
    r = fullcheck.compare_all(text, offs, ids, io, blob, chunk=1200)
    assert r["compared"] == r["sentences"] == 5000 and r["differing"] == 0 and r["first"] is None
    bad = np.array(ids, copy=True)
    io_i = np.asarray(io).astype(np.int64)
    bad[io_i[1300]] ^= 1                    # one id of sentence 1300
    bad[io_i[4000] + 1] ^= 1
    r = fullcheck.compare_all(text, offs, bad, io, blob, chunk=1200)
    assert r["differing"] == 2 and r["first"] == 1300
    io2 = np.array(io_i, copy=True)
    io2[2500:] += 1                         # sentence 2499 one id longer: every later range shifts, only 2499 changes length
    ids2 = np.insert(np.asarray(ids), io_i[2500], 7)
    r = fullcheck.compare_all(text, offs, ids2, io2.astype(np.uint64), blob, chunk=1200)
    assert r["differing"] == 1 and r["first"] == 2499
