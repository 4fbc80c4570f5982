"""Batch Decode (ids -> text).  Reference: SentencePieceProcessor::Decode(ids, &text),
src/sentencepiece_processor.cc:761-925.

 * the oracle restatement (oracle/spm_oracle.c decode_ids) against digests the COMPILED REFERENCE produced for the
   golden ids of every manifest pair and for seeded random id sequences (scripts/make_fixtures.py), and against
   the compiled reference itself where it is built (this container);
 * the device kernels (csrc/kernels_decode.h) under the emulator against the oracle;
 * -m gpu: the HIP path through the C ABI against the same digests, its error behaviour, and
   encode(decode(encode(x))) == encode(x) on a large batch."""
import hashlib
import json
import os

import numpy as np
import pytest

from scripts import make_fixtures as mf
from tests import fixtures, refshim


def _manifest():
    with open(os.path.join(fixtures.GOLDEN, "manifest.json")) as f:
        return json.load(f)


def _keys():
    return sorted(k for k in _manifest() if not k.startswith("_"))


def _digest(text, offs):
    return hashlib.sha256(np.ascontiguousarray(text).tobytes() + np.ascontiguousarray(offs).astype("<u8").tobytes()).hexdigest()


def _golden_ids(key, m, oracle, corpora):
    """The ids of a golden pair (the oracle's encode is pinned to them by tests/test_oracle.py)."""
    o = oracle.load(fixtures.model_blob(m["model"]))
    if m["options"]:
        o.set_encode_extra_options(m["options"])
    return o.encode_batch(*corpora[m["corpus"]])


@pytest.mark.parametrize("key", _keys())
def test_oracle_decode_matches_reference_digest(key, oracle, corpora):
    m = _manifest()[key]
    ids, io = _golden_ids(key, m, oracle, corpora)
    text, offs = oracle.load(fixtures.model_blob(m["model"])).decode_batch(ids, io)
    assert len(text) == m["decode_bytes"]
    assert _digest(text, offs) == m["decode_sha256"]


@pytest.mark.parametrize("model", sorted(_manifest()["_decode_fuzz"]))
def test_oracle_decode_fuzz(model, oracle):
    g = _manifest()["_decode_fuzz"][model]
    o = oracle.load(fixtures.model_blob(model))
    ids, io = mf.decode_fuzz_ids(o.lib.oracle_piece_size(o.h))
    text, offs = o.decode_batch(ids, io)
    assert len(text) == g["bytes"] and _digest(text, offs) == g["sha256"]
    if refshim.available():
        rt, ro = refshim.RefLib().load(fixtures.model_blob(model)).decode_batch(ids, io)
        np.testing.assert_array_equal(ro, offs)
        np.testing.assert_array_equal(rt, text)


def test_oracle_decode_invalid_id(oracle):
    o = oracle.load(fixtures.model_blob("test_model"))
    with pytest.raises(RuntimeError):
        o.decode_batch(np.array([5, 100000], dtype=np.int32), np.array([0, 2], dtype=np.uint64))
    with pytest.raises(RuntimeError):
        o.decode_batch(np.array([-1], dtype=np.int32), np.array([0, 1], dtype=np.uint64))


EMU_MODELS = ["test_model", "test_ja_model", "uni1k_bf", "bpe1k_bf_uds", "uni1k_ident", "uni1k_suffix", "bpe1k_noesc",
              "c5_250k_bf", "bpe1k_llama"]


@pytest.mark.parametrize("model", EMU_MODELS)
def test_emu_decode(model, oracle, corpora):
    from tests import emulib
    blob = fixtures.model_blob(model)
    e = emulib.EmuLib().load(blob)
    o = oracle.load(blob)
    for name, k in (("edge", 10 ** 6), ("botchan", 150), ("mixed2k", 60)):
        ids, io = o.encode_batch(*fixtures.head(*corpora[name], k))
        ot, oo = o.decode_batch(ids, io)
        et, eo = e.decode_batch(ids, io, grid=2)
        np.testing.assert_array_equal(eo, oo)
        np.testing.assert_array_equal(et, ot)
    ids, io = mf.decode_fuzz_ids(o.lib.oracle_piece_size(o.h))
    ot, oo = o.decode_batch(ids, io)
    et, eo = e.decode_batch(ids, io, grid=3)
    np.testing.assert_array_equal(eo, oo)
    np.testing.assert_array_equal(et, ot)
    with pytest.raises(RuntimeError):
        e.decode_batch(np.array([1, 2, 10 ** 7], dtype=np.int32), np.array([0, 1, 3], dtype=np.uint64))
    assert "Invalid id" in e.lib.spmx_last_error(None).decode()


# ------------------------------------------------------------------ GPU ----
@pytest.fixture(scope="module")
def procs():
    from sentencepiece_amd.processor import SentencePieceProcessor
    cache = {}

    def get(model):
        if model not in cache:
            cache[model] = SentencePieceProcessor(model_proto=fixtures.model_blob(model))
        return cache[model]
    return get


@pytest.mark.gpu
@pytest.mark.parametrize("key", _keys())
def test_gpu_decode_golden(key, procs, corpora):
    m = _manifest()[key]
    sp = procs(m["model"])
    sp.SetEncodeExtraOptions(m["options"])
    ids, io = sp.EncodePacked(*corpora[m["corpus"]])
    sp.SetEncodeExtraOptions("")
    text, offs = sp.DecodePacked(ids, io)
    assert len(text) == m["decode_bytes"]
    assert _digest(text, offs) == m["decode_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("model", sorted(_manifest()["_decode_fuzz"]))
def test_gpu_decode_fuzz(model, procs, oracle):
    g = _manifest()["_decode_fuzz"][model]
    sp = procs(model)
    ids, io = mf.decode_fuzz_ids(sp.GetPieceSize())
    text, offs = sp.DecodePacked(ids, io)
    assert len(text) == g["bytes"] and _digest(text, offs) == g["sha256"]
    ot, oo = oracle.load(fixtures.model_blob(model)).decode_batch(ids, io)
    np.testing.assert_array_equal(offs, oo)
    np.testing.assert_array_equal(text, ot)


@pytest.mark.gpu
def test_gpu_decode_api_and_errors(procs):
    sp = procs("test_model")
    ids = sp.Encode("I saw a girl with a telescope.")
    assert sp.Decode(ids) == "I saw a girl with a telescope."
    assert sp.Decode([ids, [], ids[:3]])[1] == ""
    assert sp.Decode([]) == ""
    with pytest.raises(Exception) as ei:
        sp.Decode([1, 2, sp.GetPieceSize()])
    assert "Invalid id" in str(ei.value)
    with pytest.raises(Exception):
        sp.Decode([[-1]])


@pytest.mark.gpu
def test_gpu_decode_round_trip_large(procs):
    """Size-independent property on 2 M sentences, device-resident: for every sentence without an unknown piece,
    encode(decode(encode(x))) == encode(x)."""
    import torch
    from sentencepiece_amd import synth
    sp = procs("uni32k")
    text, offs = synth.ascii_corpus(2_000_000, seed=4242)
    dev = torch.device("cuda", 0)
    d_ids, d_io, total = sp.EncodeDevice(torch.from_numpy(text).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev))
    d_text, d_to, nbytes = sp.DecodeDevice(d_ids[:total], d_io)
    assert nbytes > 0
    d_ids2, d_io2, total2 = sp.EncodeDevice(d_text[:nbytes], d_to)
    ids, io = d_ids[:total].cpu().numpy(), d_io.cpu().numpy()
    ids2, io2 = d_ids2[:total2].cpu().numpy(), d_io2.cpu().numpy()
    has_unk = np.add.reduceat((ids == sp.unk_id()).astype(np.int64), io[:-1].clip(max=len(ids) - 1)) > 0
    has_unk &= np.diff(io) > 0
    assert has_unk.mean() < 0.05
    cnt, cnt2 = np.diff(io), np.diff(io2)
    ok = ~has_unk
    assert np.array_equal(cnt[ok], cnt2[ok])
    keep = np.repeat(ok, cnt)
    keep2 = np.repeat(ok, cnt2)
    assert np.array_equal(ids[keep], ids2[keep2])


def test_emu_decode_follows_set_vocabulary(corpora):
    """SetVocabulary turns the pieces outside the vocabulary -- the BYTE pieces of a byte-fallback model among them --
    UNUSED (src/sentencepiece_processor.cc:301-340); Decode then no longer reassembles bytes from them: IsByte(id) reads
    the live type (:812-823).  The device's per-id decode tables are rebuilt with the types.  Against the compiled
    reference (the oracle keeps the load-time tables)."""
    from tests import emulib, refshim
    if not refshim.available():
        pytest.skip("oracle/_ref/libspm_ref.so not built")
    import sentencepiece as spm   # only to list piece strings
    blob = fixtures.model_blob("uni1k_bf")
    sp = spm.SentencePieceProcessor(model_proto=blob)
    vocab = [sp.id_to_piece(i) for i in range(300, sp.get_piece_size(), 2)]      # no <0x..> piece among them
    e, r = emulib.EmuLib().load(blob), refshim.RefLib().load(blob)
    text, offs = fixtures.head(*corpora["edge"], 60)
    ids, io = e.encode_batch(text, offs)                     # (byte pieces for the unknown characters)
    for step in ("before", "restricted", "reset"):
        if step == "restricted":
            e.set_vocabulary(vocab)
            r.set_vocabulary(vocab)
        elif step == "reset":
            e.reset_vocabulary()
            r.reset_vocabulary()
        et, eo = e.decode_batch(ids, io)
        rt, ro = r.decode_batch(ids, io)
        np.testing.assert_array_equal(eo, ro, err_msg=step)
        np.testing.assert_array_equal(et, rt, err_msg=step)


def _with_denormalizer(model, flags):
    """The model plus a denormalizer_spec: the nmt_nfkc rules (taken from test_model's normalizer_spec) as the
    'denormalization', with the three whitespace flags as given (the trainer writes False for all, sentencepiece_trainer.cc
    :145-148; the processor honours whatever the proto holds)."""
    from sentencepiece import sentencepiece_model_pb2 as pb
    m, src = pb.ModelProto(), pb.ModelProto()
    m.ParseFromString(fixtures.model_blob(model))
    src.ParseFromString(fixtures.model_blob("test_model"))
    m.denormalizer_spec.precompiled_charsmap = src.normalizer_spec.precompiled_charsmap
    m.denormalizer_spec.add_dummy_prefix = flags[0]
    m.denormalizer_spec.remove_extra_whitespaces = flags[1]
    m.denormalizer_spec.escape_whitespaces = flags[2]
    return m.SerializeToString()


def _denormalizer_case(load, corpora):
    """Decode of a model with a denormalizer_spec: `*text = denormalizer_->Normalize(*text)` (src/sentencepiece_processor.cc
    :905-907) -- against the compiled reference (the oracle does not restate the denormalizer)."""
    r = refshim.RefLib()
    for model, flags in (("uni1k_ident", (False, False, False)), ("uni1k_ident", (True, True, True)),
                         ("uni1k_bf", (False, True, False)), ("bpe1k_noesc", (False, False, False))):
        blob = _with_denormalizer(model, flags)
        e, rh = load(blob), r.load(blob)
        for name, k in (("edge", 10 ** 6), ("mixed2k", 80), ("ja", 20)):
            text, offs = fixtures.head(*corpora[name], k)
            ids, io = rh.encode_batch(text, offs)
            rt, ro = rh.decode_batch(ids, io)
            et, eo = e.DecodePacked(ids, io)
            np.testing.assert_array_equal(eo, ro, err_msg=str((model, flags, name)))
            np.testing.assert_array_equal(et, rt, err_msg=str((model, flags, name)))
        # the identity-normalized models keep full-width / ligature / circled characters in their ids (as UNK or bytes or
        # pieces); what the denormalizer sees includes the unk surface U+2047, which nmt_nfkc rewrites to "??"
        assert b"\xe2\x81\x87" not in bytes(et) or flags is None


def test_emu_decode_with_denormalizer(corpora):
    from tests import emulib
    if not refshim.available():
        pytest.skip("oracle/_ref/libspm_ref.so not built")
    lib = emulib.EmuLib()
    _denormalizer_case(lambda blob: lib.load(blob).sp, corpora)


@pytest.mark.gpu
def test_gpu_decode_with_denormalizer(corpora):
    from sentencepiece_amd.processor import SentencePieceProcessor
    if not refshim.available():
        pytest.skip("oracle/_ref/libspm_ref.so not built")
    _denormalizer_case(lambda blob: SentencePieceProcessor(model_proto=blob), corpora)


# ---- SetDecodeExtraOptions (src/sentencepiece_processor.h:270, .cc:288-291, applied at :819) --------------------------
DECODE_OPTION_MODELS = ["test_model", "uni1k_bf", "bpe1k_llama"]
DECODE_OPTIONS = ["reverse", "bos:eos", "unk", "eos:reverse:bos", "reverse:reverse:bos", ""]


def _decode_option_case(sp, model, oracle, corpora):
    """sp: the product's processor (HIP or emulated).  Its Decode under every option string equals the compiled reference's."""
    from tests import refshim
    blob = fixtures.model_blob(model)
    o = oracle.load(blob)
    r = refshim.RefLib().load(blob) if refshim.available() else None
    batches = [o.encode_batch(*fixtures.head(*corpora[name], k)) for name, k in (("edge", 10 ** 6), ("botchan", 120), ("mixed2k", 50))]
    batches.append(mf.decode_fuzz_ids(o.lib.oracle_piece_size(o.h)))
    try:
        for opts in DECODE_OPTIONS:
            sp.SetDecodeExtraOptions(opts)
            if r is not None:
                r.set_decode_extra_options(opts)
            for ids, io in batches:
                text, offs = sp.DecodePacked(ids, io)
                if r is not None:
                    rt, ro = r.decode_batch(ids, io)
                    np.testing.assert_array_equal(offs, ro, err_msg="%s %r" % (model, opts))
                    np.testing.assert_array_equal(text, rt, err_msg="%s %r" % (model, opts))
                if opts in ("", "unk", "bos:eos"):      # (control pieces decode to nothing; `unk` only renames pieces)
                    ot, oo = o.decode_batch(ids, io)
                    np.testing.assert_array_equal(offs, oo)
                    np.testing.assert_array_equal(text, ot)
        with pytest.raises(Exception) as ei:
            sp.SetDecodeExtraOptions("bos:nonsense")
        assert "not available" in str(ei.value)
    finally:
        sp.SetDecodeExtraOptions("")


@pytest.mark.parametrize("model", DECODE_OPTION_MODELS)
def test_emu_decode_extra_options(model, oracle, corpora):
    from tests import emulib
    _decode_option_case(emulib.EmuLib().load(fixtures.model_blob(model)).sp, model, oracle, corpora)


@pytest.mark.gpu
@pytest.mark.parametrize("model", DECODE_OPTION_MODELS)
def test_gpu_decode_extra_options(model, procs, oracle, corpora):
    _decode_option_case(procs(model), model, oracle, corpora)
    sp = procs(model)                       # the string forms go through the same entry point
    sp.SetDecodeExtraOptions("reverse")
    try:
        ids = sp.Encode("I saw a girl with a telescope.")
        got = sp.Decode(ids)
        sp.SetDecodeExtraOptions("")
        assert got == sp.Decode(list(reversed(ids)))
    finally:
        sp.SetDecodeExtraOptions("")


# ---------------------------------------------------------- Decode(pieces) ----
PIECE_MODELS = ["test_model", "uni1k_bf", "bpe1k_bf_uds", "uni1k_suffix", "bpe1k_llama", "test_ja_model"]


def _piece_rows(ref, rng):
    """Piece lists the reference itself produced, then damaged: pieces that are in no vocabulary (plain, with a space
    symbol in front, empty, very long), the unknown piece by name, byte pieces, control pieces."""
    texts = ["I saw a girl with a telescope.", "  Hello   world  ", "吾輩は猫である。", "tab\there 　x", "", "a", "€uro ＡＢ"]
    rows = [ref.encode_pieces_one(t) for t in texts]
    extra = ["zzzqqq", "▁notapiece", "", "<unk>", "<s>", "</s>", "<0xE3>", "<0x81>", "<0x82>", "<0xFF>", "▁", "x" * 300,
             "▁▁double", "日本語のかたまり"]
    out = list(rows)
    for r in rows:
        for _ in range(3):
            d = list(r)
            for _ in range(int(rng.integers(1, 4))):
                d.insert(int(rng.integers(0, len(d) + 1)), extra[int(rng.integers(0, len(extra)))])
            out.append(d)
    out.append(["zzzqqq"])                      # nothing but a piece outside the vocabulary
    out.append(["▁notapiece", "▁the"])
    out.append(["<0xE3>", "<0x81>", "zz", "<0x82>"])     # a byte run cut by a literal
    return out


def _check_decode_pieces(sp, ref, rng):
    rows = _piece_rows(ref, rng)
    for opts in ("", "reverse", "bos:eos", "unk", "eos:reverse:unk_piece"):
        sp.SetDecodeExtraOptions(opts)
        ref.set_decode_extra_options(opts)
        want = [ref.decode_pieces(r) for r in rows]
        got = sp.DecodePieces(rows, out_type=bytes)
        assert got == want, (opts, [(r, g, w) for r, g, w in zip(rows, got, want) if g != w][:3])
        assert sp.DecodePieces(rows[-1], out_type=bytes) == want[-1]         # the single-sentence form
    sp.SetDecodeExtraOptions("")
    ref.set_decode_extra_options("")


def _ref_with_pieces(model):
    ref = refshim.RefLib().load(fixtures.model_blob(model))

    def one(t):
        b = np.frombuffer(t.encode("utf-8"), dtype=np.uint8)
        _, _, _, _, blob, po = ref.encode_pieces(b, np.array([0, len(b)], dtype=np.uint64))
        return [blob[int(po[k]):int(po[k + 1])].decode("utf-8", "surrogateescape") for k in range(len(po) - 1)]
    ref.encode_pieces_one = one
    return ref


@pytest.mark.skipif(not refshim.available(), reason="the compiled reference (oracle/_ref) is not built")
@pytest.mark.parametrize("model", PIECE_MODELS)
def test_emu_decode_pieces(model):
    """Decode(const std::vector<std::string>& pieces, ...) (src/sentencepiece_processor.cc:761-769): a piece that is not in
    the vocabulary is copied through as text (:784-790), the unknown piece by name becomes unk_surface, byte pieces are
    reassembled across it, the decode extra options apply to the piece list first -- against the compiled reference."""
    from tests import emulib
    e = emulib.EmuLib().load(fixtures.model_blob(model))
    _check_decode_pieces(e.sp, _ref_with_pieces(model), np.random.default_rng(7))


@pytest.mark.gpu
@pytest.mark.skipif(not refshim.available(), reason="the compiled reference (oracle/_ref) is not built")
@pytest.mark.parametrize("model", PIECE_MODELS)
def test_gpu_decode_pieces(model, procs):
    _check_decode_pieces(procs(model), _ref_with_pieces(model), np.random.default_rng(7))
