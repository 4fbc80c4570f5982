"""The product's device bodies (sentencepiece_amd/csrc/kernels*.h) and host
table compiler run on the CPU under the lock-step wavefront model of
tests/emu/ and are compared with the oracle.  Small inputs only (the model
executes 64 fibers per wave); the full-size parity runs are the -m gpu tests."""
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
        ids, io = h.encode_batch(text, offs, grid=3)
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
        ids, io = h.encode_batch(tx, ox, grid=2)
        assert h.status == 0
        oids, oio = o.encode_batch(tx, ox)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)


@pytest.mark.parametrize("model", ["test_model", "uni1k_bf", "uni1k_suffix", "uni1k_ident", "uni32k"])
@pytest.mark.parametrize("env", [{}, {"SPMX_NO_COMPRESS": "1"}, {"SPMX_NO_FAST": "1"},
                                 {"SPMX_NO_COMPRESS": "1", "SPMX_NO_FAST": "1"}])
def test_emu_tile_variants(model, env, emu, oracle, corpora, monkeypatch):
    """One-byte space symbol on/off x FAST per-lane normalizer on/off: same ids; with both on, ASCII sentences
    stay in the FAST kernel and the rest is handed over."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    blob = fixtures.model_blob(model)
    h = emu.load(blob)
    assert bool(h.flags() & K_COMPRESS) == ("SPMX_NO_COMPRESS" not in env)
    o = oracle.load(blob)
    for name, k in (("edge", 10 ** 6), ("synth20k", 300), ("mixed2k", 40), ("botchan", 150)):
        text, offs = fixtures.head(*corpora[name], k)
        ids, io = h.encode_batch(text, offs, grid=2)
        assert h.status == 0
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
        kept, handed = h.fast_split()
        if not env and name == "synth20k" and model != "uni1k_suffix":   # suffix mode: GENERAL kernel only
            assert kept > 0.9 * (len(offs) - 1)
        if "SPMX_NO_FAST" in env or "SPMX_NO_COMPRESS" in env:
            assert kept == 0


K_WORDWISE = 1 << 10   # dev.h kNfBpeWordwise


@pytest.mark.parametrize("model,corpus,k", [("test_ja_model", "ja", 200), ("c5_250k", "mixed2k", 200),
                                             ("uni1k_bf", "edge", 10 ** 6), ("bpe32k", "mixed2k", 40)])
@pytest.mark.parametrize("env", [{}, {"SPMX_NO_LANE_GENERAL": "1"}])
def test_emu_lane_general_normalizer(model, corpus, k, env, emu, oracle, corpora, monkeypatch):
    """Non-ASCII text: the per-lane general normalizer keeps the sentences in the FAST kernel (charsmap rules,
    malformed UTF-8, literal U+2581 ...); with it switched off they go through normalize_wave.  Same ids."""
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    blob = fixtures.model_blob(model)
    h = emu.load(blob)
    o = oracle.load(blob)
    text, offs = fixtures.head(*corpora[corpus], k)
    ids, io = h.encode_batch(text, offs, grid=2)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    kept, handed = h.fast_split()
    if env:      # (characters that start no charsmap key stay in the FAST kernel's own normalizer either way)
        assert handed > 0.25 * (len(offs) - 1)
    elif model != "uni1k_bf":      # (the edge cases are mostly ASCII tiles: stray non-ASCII sentences are handed over)
        assert kept > 0.3 * (len(offs) - 1)


@pytest.mark.parametrize("model", ["bpe1k", "bpe32k", "bpe1k_bf_uds", "bpe1k_noesc", "bpe1k_llama"])
@pytest.mark.parametrize("env", [{}, {"SPMX_NO_WORDWISE": "1"}, {"SPMX_NO_COMPRESS": "1"}, {"SPMX_NO_FAST": "1"},
                                 {"SPMX_NO_STREAM": "1"}])
def test_emu_bpe_variants(model, env, emu, oracle, corpora, monkeypatch):
    """BPE: lane-per-sentence word-by-word form (word-wise models) vs sentence-per-wave form: same ids; long
    words are handed to the sentence-per-wave kernel."""
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
        ids, io = h.encode_batch(text, offs, grid=2)
        assert h.status == 0
        oids, oio = o.encode_batch(text, offs)
        np.testing.assert_array_equal(io, oio)
        np.testing.assert_array_equal(ids, oids)
        if name == "edge" and wordwise and "SPMX_NO_STREAM" not in env:
            assert h.wave_handed() >= 3      # "a" * 300, "0123456789" * 40, a long CJK run ...
        if name == "synth20k" and wordwise and not env:
            assert h.fast_split()[0] > 0.9 * (len(offs) - 1) and h.wave_handed() == 0


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
    st = C.c_uint32(0)
    tot = h.lib.emu_encode_batch(h.h, np.ascontiguousarray(text).ctypes.data, np.ascontiguousarray(offs).ctypes.data,
                                 len(offs) - 1, ids.ctypes.data, 8, id_offs.ctypes.data, 2, C.byref(st))
    assert tot == -len(full) - 2
    assert (ids == -7).all()
    np.testing.assert_array_equal(id_offs, io)


@pytest.mark.parametrize("model", ["test_model", "c5_250k_bf", "test_ja_model"])
def test_emu_document_length(model, emu, oracle, corpora):
    """Sentences beyond the staged classes (> 8192 B raw): the FAST kernel alone, per-lane normalizers for ASCII and
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
    ids, io = h.encode_batch(text, offs, grid=2)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)


def test_emu_document_length_needs_fast_model(emu, corpora):
    """A model the per-lane normalizers cannot take (user-defined symbols) is limited to the staged classes."""
    from sentencepiece_amd import synth
    h = emu.load(fixtures.model_blob("uni1k_uds"))
    text, offs = synth.pack([b"y" * 9000])
    h.encode_batch(text, offs, grid=1)
    assert h.status & 2          # kStTooLong: csrc/api.cc turns it into OUT_OF_RANGE


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
    ids, io = h.encode_batch(text, offs, grid=2)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    kept, handed = h.fast_split()
    assert kept >= 0.7 * 300 and handed >= n_leave // 2     # (a stray byte can start a key together with its neighbour)


@pytest.mark.parametrize("model", ["bpe1k", "bpe1k_llama"])
def test_emu_bpe_documents(model, emu, oracle):
    """BPE document-length class: the lane form with the HBM merge for words that outgrow the LDS slots."""
    from tests.test_gpu_parity import bpe_documents
    text, offs = bpe_documents()
    text, offs = fixtures.head(text, offs, 8)
    blob = fixtures.model_blob(model)
    h, o = emu.load(blob), oracle.load(blob)
    ids, io = h.encode_batch(text, offs, grid=2)
    assert h.status == 0
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
