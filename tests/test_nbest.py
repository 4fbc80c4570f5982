"""NBestEncode -- the ORACLE only, pinned to the compiled reference: the lattice
(Lattice::SetSentence / PopulateNodes), the float Viterbi and the A* of Lattice::NBest with libstdc++'s heap order.
The device kernel for it does not exist yet; this is the checker it will be held to."""
import ctypes as C

import numpy as np
import pytest

from tests import fixtures

UNIGRAM = ["test_model", "test_ja_model", "uni1k", "uni1k_bf", "uni1k_uds", "uni1k_ident", "uni1k_suffix", "uni32k"]
ARG = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]


def nbest(fn, h, text, k):
    fn.restype = C.c_int64
    fn.argtypes = ARG
    cap = (len(text) + 8) * 4 * max(k, 1) + 64
    ids = np.empty(cap, dtype=np.int32)
    offs = np.zeros(min(max(k, 1), 1024) + 2, dtype=np.uint64)
    scores = np.zeros(min(max(k, 1), 1024) + 1, dtype=np.float32)
    n = fn(h, text, len(text), k, ids.ctypes.data, cap, offs.ctypes.data, scores.ctypes.data)
    if n < 0:
        return n, None, None
    return n, [ids[int(offs[i]):int(offs[i + 1])].tolist() for i in range(n)], scores[:n].copy()


@pytest.fixture(scope="module")
def ref():
    from tests import refshim
    if not refshim.available():
        pytest.skip("oracle/_ref/libspm_ref.so not built")
    return refshim.RefLib()


def sentences(corpora):
    out = [b"", b" ", b"a", b"hello world", b"aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa", ("吾輩は猫である" * 3).encode(), b"\xff\xfe", "ＡＢＣ ① ㍿".encode(),
           b"the the the the the the the the the the the the the the the the the the the the the the the the the"]
    for name, k in (("edge", 10 ** 6), ("botchan", 60), ("ja", 8), ("mixed2k", 10)):
        text, offs = fixtures.head(*corpora[name], k)
        tb = np.asarray(text).tobytes()
        out += [tb[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    return [s[:600] for s in out]


@pytest.mark.parametrize("model", UNIGRAM)
def test_oracle_nbest_matches_reference(model, oracle, ref, corpora):
    blob = fixtures.model_blob(model)
    o, r = oracle.load(blob), ref.load(blob)
    for opts in ("", "bos:eos"):
        o.set_encode_extra_options(opts)
        r.set_encode_extra_options(opts)
        for s in sentences(corpora):
            for k in (1, 2, 5, 17):
                n1, a, sa = nbest(o.lib.oracle_nbest_encode, o.h, s, k)
                n2, b, sb = nbest(r.lib.spmref_nbest_encode, r.h, s, k)
                assert n1 == n2, (model, s[:40], k)
                assert a == b, (model, s[:40], k)
                np.testing.assert_array_equal(sa, sb)


def test_oracle_nbest_agenda_shrink(oracle, ref):
    """A long ambiguous input drives the agenda past 10000 entries: the shrink to min(512, 10 nbest) (:487-514)."""
    blob = fixtures.model_blob("test_model")
    o, r = oracle.load(blob), ref.load(blob)
    s = (b"this is a test of the emergency broadcast system " * 6)[:280]
    for k in (64, 400):
        n1, a, sa = nbest(o.lib.oracle_nbest_encode, o.h, s, k)
        n2, b, sb = nbest(r.lib.spmref_nbest_encode, r.h, s, k)
        assert n1 == n2 and a == b
        np.testing.assert_array_equal(sa, sb)


def test_nbest_is_unigram_only(oracle, ref):
    blob = fixtures.model_blob("bpe1k")
    o, r = oracle.load(blob), ref.load(blob)
    assert nbest(o.lib.oracle_nbest_encode, o.h, b"hello", 3)[0] == -1
    assert nbest(r.lib.spmref_nbest_encode, r.h, b"hello", 3)[0] == -1


@pytest.fixture(scope="module")
def emu():
    from tests import emulib
    return emulib.EmuLib()


@pytest.mark.parametrize("model", UNIGRAM)
def test_emu_nbest_matches_oracle(model, emu, oracle, corpora):
    """The device NBest kernel (kernels_nbest.h, emulated) against the oracle: ids and scores of every result."""
    from sentencepiece_amd import synth
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob), oracle.load(blob)
    sents = [s for s in sentences(corpora) if len(s) <= 250]
    text, offs = synth.pack(sents)
    for opts in ("", "reverse:bos:eos"):
        h.set_encode_extra_options(opts)
        o.set_encode_extra_options(opts)
        for k in (2, 5, 17):
            got = h.nbest(text, offs, k)
            assert h.status == 0
            for s, res in zip(sents, got):
                n, want, sc = nbest(o.lib.oracle_nbest_encode, o.h, s, k)
                assert [r[0] for r in res] == want, (model, s[:40], k, opts)
                np.testing.assert_array_equal(np.array([r[1] for r in res], dtype=np.float32), sc)


def test_emu_nbest_agenda_shrink_and_limits(emu, oracle):
    from sentencepiece_amd import synth
    blob = fixtures.model_blob("test_model")
    h, o = emu.load(blob), oracle.load(blob)
    s = (b"this is a test of the emergency broadcast system " * 6)[:280]
    text, offs = synth.pack([s, b"", b"a"])
    for k in (64, 400):
        got = h.nbest(text, offs, k, grid=1)
        for sent, res in zip([s, b"", b"a"], got):
            n, want, sc = nbest(o.lib.oracle_nbest_encode, o.h, sent, k)
            assert [r[0] for r in res] == want
            np.testing.assert_array_equal(np.array([r[1] for r in res], dtype=np.float32), sc)
    # beyond the first launch's capacities (1024 normalized bytes / 16384 nodes): the wide launch, no length limit
    long_sents = [b"word " * 400, b"a", (b"the quick brown fox jumps over the lazy dog " * 300)[:12000], s]
    got = h.nbest(*synth.pack(long_sents), 4, grid=1)
    for sent, res in zip(long_sents, got):
        n, want, sc = nbest(o.lib.oracle_nbest_encode, o.h, sent, 4)
        assert [r[0] for r in res] == want
        np.testing.assert_array_equal(np.array([r[1] for r in res], dtype=np.float32), sc)


# The device path on hardware (through the C ABI).
@pytest.mark.gpu
@pytest.mark.parametrize("model", ["test_model", "uni1k_bf", "uni32k"])
def test_gpu_nbest_matches_oracle(model, oracle, corpora):
    from sentencepiece_amd import synth
    from sentencepiece_amd.processor import SentencePieceProcessor
    blob = fixtures.model_blob(model)
    sp, o = SentencePieceProcessor(model_proto=blob), oracle.load(blob)
    sents = [s for s in sentences(corpora) if len(s) <= 250]
    text, offs = synth.pack(sents)
    for opts in ("", "reverse:bos:eos"):
        sp.SetEncodeExtraOptions(opts)
        o.set_encode_extra_options(opts)
        for k in (1, 2, 5, 17):
            ids, io, sc, ro = sp.NBestPacked(text, offs, k)
            for i, s in enumerate(sents):
                n, want, wsc = nbest(o.lib.oracle_nbest_encode, o.h, s, k)
                got = [ids[int(io[r]):int(io[r + 1])].tolist() for r in range(int(ro[i]), int(ro[i + 1]))]
                assert got == want, (model, s[:40], k, opts)
                np.testing.assert_array_equal(sc[int(ro[i]):int(ro[i + 1])], wsc)
    assert sp.NBestEncodeAsIds("hello world", 3) == nbest(o.lib.oracle_nbest_encode, o.h, b"hello world", 3)[1]


def test_nbest_with_restricted_vocabulary(emu, oracle, ref, corpora):
    """SetVocabulary marks pieces UNUSED; PopulateNodes skips them (src/unigram_model.cc:576)."""
    from sentencepiece import sentencepiece_model_pb2 as pb
    from sentencepiece_amd import synth
    blob = fixtures.model_blob("test_model")
    m = pb.ModelProto()
    m.ParseFromString(blob)
    keep = [p.piece for i, p in enumerate(m.pieces) if i % 3 != 1]
    h, o, r = emu.load(blob), oracle.load(blob), ref.load(blob)
    for x in (h, o, r):
        x.set_vocabulary(keep)
    sents = [s for s in sentences(corpora) if 0 < len(s) <= 200][:60]
    got = h.nbest(*synth.pack(sents), 6)
    for s, res in zip(sents, got):
        n1, a, sa = nbest(o.lib.oracle_nbest_encode, o.h, s, 6)
        n2, b, sb = nbest(r.lib.spmref_nbest_encode, r.h, s, 6)
        assert a == b and [x[0] for x in res] == a
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(np.array([x[1] for x in res], dtype=np.float32), sa)


# A batch large enough for the release library's threaded result assembly (api.cc LatticeBatchHost: from 2^20 ids +
# results on; pinned ids from 8 MB on): a few hundred distinct sentences, each with its oracle answer, repeated in a
# shuffled order.
@pytest.mark.gpu
def test_gpu_nbest_big_batch_threaded_assembly(oracle, corpora):
    from sentencepiece_amd import synth
    from sentencepiece_amd.processor import SentencePieceProcessor
    blob = fixtures.model_blob("uni32k")
    sp, o = SentencePieceProcessor(model_proto=blob), oracle.load(blob)
    t0, o0 = synth.ascii_corpus(400, seed=11)
    base = [t0[int(o0[i]):int(o0[i + 1])].tobytes() for i in range(400)] + [b"", b"hello world"]
    k = 5
    want = [nbest(o.lib.oracle_nbest_encode, o.h, s, k) for s in base]
    rng = np.random.default_rng(5)
    order = rng.integers(0, len(base), size=60000)
    text, offs = synth.pack([base[j] for j in order])
    for _ in range(2):                                            # (the second call takes the recycled pinned block)
        ids, io, sc, ro = sp.NBestPacked(text, offs, k)
        assert int(ro[-1]) >= (1 << 17) and int(io[-1]) >= (1 << 20)
        for i, j in enumerate(order):
            n, w_ids, w_sc = want[j]
            r0, r1 = int(ro[i]), int(ro[i + 1])
            assert r1 - r0 == n, (i, j)
            for r in range(r0, r1):
                assert ids[int(io[r]):int(io[r + 1])].tolist() == w_ids[r - r0], (i, j, r - r0)
            np.testing.assert_array_equal(sc[r0:r1], w_sc)
        del ids, io, sc, ro


# The first launch's capacities follow the batch (api.cc LatticeBatchHost: normalized length guessed from the longest raw
# sentence, 512 hypotheses a result): a batch of short sentences that normalize to several times their length, and a
# sentence whose A* outgrows the first hypothesis slice, are set aside and answered by the wide launch.
def check_capacity_guesses(make_sp, oracle):
    from sentencepiece_amd import synth
    blob = fixtures.model_blob("test_model")                      # nmt_nfkc: U+3300 -> four katakana, U+FDFA -> 18 code points
    sp, o = make_sp(blob), oracle.load(blob)
    grow = ["㌀" * 14, "ﷺ" * 6 + " a", "a", ""]          # 42 raw bytes -> 168 normalized; the guess is 64
    wide = [(b"this is a test of the emergency broadcast system " * 6)[:280], b"hello world"]
    for sents, k in ((grow, 3), (wide, 40), (wide + [s.encode("utf-8") for s in grow], 7)):
        sents = [s.encode("utf-8") if isinstance(s, str) else s for s in sents]
        text, offs = synth.pack(sents)
        ids, io, sc, ro = sp.NBestPacked(text, offs, k)
        for i, s in enumerate(sents):
            n, want, wsc = nbest(o.lib.oracle_nbest_encode, o.h, s, k)
            got = [ids[int(io[r]):int(io[r + 1])].tolist() for r in range(int(ro[i]), int(ro[i + 1]))]
            assert got == want, (s[:30], k)
            np.testing.assert_array_equal(sc[int(ro[i]):int(ro[i + 1])], wsc)


def test_emu_nbest_capacity_guesses(emu, oracle):
    check_capacity_guesses(lambda blob: emu.load(blob).sp, oracle)


@pytest.mark.gpu
def test_gpu_nbest_capacity_guesses(oracle):
    from sentencepiece_amd.processor import SentencePieceProcessor
    check_capacity_guesses(lambda blob: SentencePieceProcessor(model_proto=blob), oracle)
