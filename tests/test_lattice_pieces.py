"""Piece / proto forms of NBestEncode and SampleEncode (src/sentencepiece_processor.h:318-324,
:346-353, :404-408, .cc:653-720): spmx_nbest_encode_batch_spans / spmx_sample_encode_batch_spans and the Python entry
points over them.

  * golden: the serialized NBestSentencePieceText of the REFERENCE (tests/golden/nbest_protos.json, made by
    scripts/make_nbest_golden.py) -- pieces, ids, surfaces, byte ranges and scores of every result;
  * extra options (bos / eos / reverse / unk_piece): n-best with nbest_size 1 and SampleEncode with nbest_size 1 are
    the plain encoder's SentencePieceText, which has reference-made vectors of its own (tests/test_pieces.py);
  * every drawn segmentation tiles the normalized text and the input.
CPU: the product's api.cc + kernels under the emulator; -m gpu: the HIP path."""
import json
import os
import struct

import numpy as np
import pytest

from tests import fixtures

GOLD = os.path.join(fixtures.GOLDEN, "nbest_protos.json")


@pytest.fixture(scope="module")
def emu():
    from tests import emulib
    return emulib.EmuLib()


def parse_spt(blob):
    """SentencePieceText wire format -> (text, [(piece, id, surface | None, begin, end)], score | None)."""
    def varint(b, i):
        v = s = 0
        while True:
            c = b[i]; i += 1
            v |= (c & 0x7F) << s; s += 7
            if not c & 0x80:
                return v, i
    def fields(b):
        i = 0
        while i < len(b):
            tag, i = varint(b, i)
            num, wt = tag >> 3, tag & 7
            if wt == 0:
                v, i = varint(b, i)
            elif wt == 2:
                ln, i = varint(b, i); v = b[i:i + ln]; i += ln
            elif wt == 5:
                v = struct.unpack("<f", b[i:i + 4])[0]; i += 4
            else:
                raise AssertionError("wire type %d" % wt)
            yield num, v
    text, pieces, score = b"", [], None
    for num, v in fields(blob):
        if num == 1:
            text = v
        elif num == 2:
            d = {"piece": b"", "id": 0, "surface": None, "begin": 0, "end": 0}
            for n2, v2 in fields(v):
                d[{1: "piece", 2: "id", 3: "surface", 4: "begin", 5: "end"}[n2]] = v2
            pieces.append((d["piece"], d["id"], d["surface"], d["begin"], d["end"]))
        elif num == 3:
            score = v
    return text, pieces, score


def parse_nbest(blob):
    out, i = [], 0
    while i < len(blob):
        assert blob[i] == 0x0A
        i += 1
        ln = s = 0
        while True:
            c = blob[i]; i += 1
            ln |= (c & 0x7F) << s; s += 7
            if not c & 0x80:
                break
        out.append(parse_spt(blob[i:i + ln]))
        i += ln
    return out


def check_golden(make_sp):
    with open(GOLD, encoding="utf-8") as f:
        gold = json.load(f)
    sents = gold["sentences"]
    sps = {}
    for case in gold["cases"]:
        m = case["model"]
        if m not in sps:
            sps[m] = make_sp(fixtures.model_blob(m))
        sp = sps[m]
        got = sp.NBestEncodeAsSerializedProto(sents, case["nbest"])
        pieces = sp.NBestEncodeAsPieces(sents, case["nbest"])
        for k, s in enumerate(sents):
            want = parse_nbest(bytes.fromhex(case["protos"][k]))
            have = parse_nbest(got[k])
            assert len(have) == len(want), (m, case["nbest"], s)
            for (wt, wp, ws), (ht, hp, hs) in zip(want, have):
                assert ht == wt and hp == wp, (m, case["nbest"], s, hp, wp)
                assert hs is not None and abs(hs - ws) <= 1e-5 * max(1.0, abs(ws)), (m, s, hs, ws)
            assert pieces[k] == [[p.decode("utf-8", "surrogateescape") for p, *_ in wp] for _, wp, _ in want]


def check_options_and_tiling(make_sp, corpora, n_sent=40):
    text, offs = fixtures.head(*corpora["edge"], n_sent)
    b = text.tobytes()
    sents = [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    for model, opts in (("test_model", "bos:eos"), ("test_model", "reverse"), ("uni1k_bf", "eos:reverse:unk_piece"),
                        ("uni1k_bf", "reverse:bos:eos"), ("test_ja_model", "bos")):
        sp = make_sp(fixtures.model_blob(model))
        sp.SetEncodeExtraOptions(opts)
        plain = sp.EncodeAsSerializedProto(sents)
        nb1 = sp.NBestEncodeAsSerializedProto(sents, 1)
        sm1 = sp.SampleEncodeAsSerializedProto(sents, 1, 0.5)
        nb4 = sp.NBestEncodeAsSentencePieceText(sents, 4)
        ids4 = sp.NBestEncodeAsIds(sents, 4)
        draws = sp.SampleEncodeAsSentencePieceText(sents, -1, 0.3, seed=7)
        draws8 = sp.SampleEncodeAsSentencePieceText(sents, 8, 0.3, seed=7)
        norm, no, _ = sp.NormalizePacked(text, offs)
        norm = norm.tobytes()
        for i, s in enumerate(sents):
            t0, p0, _ = parse_spt(plain[i])
            (t1, p1, sc1), = parse_nbest(nb1[i])
            assert (t1, p1) == (t0, p0) and sc1 == 0.0                  # unigram_model.cc:694-696
            assert parse_spt(sm1[i])[:2] == (t0, p0)
            assert [[t for _, t, *_ in rows] for _, rows in nb4[i]] == ids4[i]
            nrm = norm[int(no[i]):int(no[i + 1])]
            for rows in [r for _, r in nb4[i]] + [draws[i], draws8[i]]:
                body = [r for r in rows if not sp.IsControl(r[1])]
                if "reverse" in opts.split(":"):
                    body = body[::-1]
                # the surfaces tile the input from the first consumed byte on; pieces that are text tile the normalized text
                pos = None
                for piece, t, surf, pb, pe in body:
                    assert pb <= pe <= len(s)
                    if surf is not None:
                        assert surf == s[pb:pe]
                    if pos is not None and not sp.IsByte(t):
                        assert pb == pos, (model, opts, s, rows)
                    if not sp.IsByte(t) or surf is not None:
                        pos = pe
                if "unk_piece" not in opts and not any(sp.IsByte(t) for _, t, *_ in body):
                    assert b"".join(p for p, *_ in body) == nrm
                for piece, t, surf, pb, pe in rows:
                    if sp.IsControl(t):
                        assert pb == pe and pb in (0, len(s)) and surf is None


def check_bpe_sample_pieces(make_sp, corpora):
    """BPE: SampleEncode is BPE-dropout for every nbest_size; alpha = 0 is the plain encoder's SentencePieceText."""
    text, offs = fixtures.head(*corpora["botchan"], 30)
    b = text.tobytes()
    sents = [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    sp = make_sp(fixtures.model_blob("bpe1k"))
    assert sp.SampleEncodeAsSerializedProto(sents, 5, 0.0) == sp.EncodeAsSerializedProto(sents)
    rows = sp.SampleEncodeAsSentencePieceText(sents, -1, 0.4, seed=3)
    norm, no, _ = sp.NormalizePacked(text, offs)
    norm = norm.tobytes()
    changed = 0
    plain = sp.EncodeAsSentencePieceText(sents)
    for i, s in enumerate(sents):
        assert b"".join(p for p, *_ in rows[i]) == norm[int(no[i]):int(no[i + 1])]
        for piece, t, sf, pb, pe in rows[i]:
            assert sf == s[pb:pe]
        changed += [t for _, t, *_ in rows[i]] != [t for _, t, *_ in plain[i]]
    assert changed > 0
    with pytest.raises(RuntimeError):
        sp.NBestEncodeAsPieces(sents[:2], 3)          # "NBestEncode is not available for the current model."


def test_emu_nbest_protos_match_the_reference(emu):
    check_golden(lambda blob: emu.load(blob).sp)


def test_emu_lattice_piece_forms_options_and_tiling(emu, corpora):
    check_options_and_tiling(lambda blob: emu.load(blob).sp, corpora, n_sent=25)


def test_emu_bpe_sample_pieces(emu, corpora):
    check_bpe_sample_pieces(lambda blob: emu.load(blob).sp, corpora)


@pytest.mark.gpu
def test_gpu_nbest_protos_match_the_reference():
    from sentencepiece_amd.processor import SentencePieceProcessor
    check_golden(lambda blob: SentencePieceProcessor(model_proto=blob))


@pytest.mark.gpu
def test_gpu_lattice_piece_forms_options_and_tiling(corpora):
    from sentencepiece_amd.processor import SentencePieceProcessor
    check_options_and_tiling(lambda blob: SentencePieceProcessor(model_proto=blob), corpora, n_sent=120)
    check_bpe_sample_pieces(lambda blob: SentencePieceProcessor(model_proto=blob), corpora)


def check_python_encode_sampling_out_types(make_sp):
    """encode(enable_sampling=True, out_type=str / "serialized_proto" / "immutable_proto") (python/src/sentencepiece/
    __init__.py Encode -> _SampleEncodeAsPieces / ...Proto): nbest_size 1 is the plain encoder's answer; drawn pieces
    spell the normalized text; add_bos / add_eos / reverse apply to the piece form and are refused by the proto forms."""
    sp = make_sp(fixtures.model_blob("test_model"))
    s = "Hello world, this is a test."
    assert sp.Encode(s, out_type=str, enable_sampling=True, nbest_size=1, alpha=0.5) == sp.Encode(s, out_type=str)
    assert (sp.Encode([s, "x y"], out_type=str, enable_sampling=True, nbest_size=1, alpha=0.5, add_bos=True, add_eos=True, reverse=True)
            == sp.Encode([s, "x y"], out_type=str, add_bos=True, add_eos=True, reverse=True))
    assert sp.Encode(s, out_type="serialized_proto", enable_sampling=True, nbest_size=1, alpha=0.5) == sp.Encode(s, out_type="serialized_proto")
    drawn = sp.Encode(s, out_type=str, enable_sampling=True, nbest_size=-1, alpha=0.2)
    assert "".join(drawn) == "".join(sp.Encode(s, out_type=str))
    imm = sp.Encode(s, out_type="immutable_proto", enable_sampling=True, nbest_size=-1, alpha=0.2)
    assert imm.text == s and "".join(p.surface for p in imm.pieces) == s and len(parse_spt(imm.SerializeAsString())[1]) == len(imm.pieces)
    with pytest.raises(NotImplementedError):
        sp.Encode(s, out_type="serialized_proto", enable_sampling=True, nbest_size=-1, alpha=0.2, add_bos=True)
    assert sp.Encode(s, out_type=str) == sp.EncodeAsPieces(s)            # (the options of the sampling calls did not stick)


def test_emu_python_encode_sampling_out_types(emu):
    check_python_encode_sampling_out_types(lambda blob: emu.load(blob).sp)


@pytest.mark.gpu
def test_gpu_python_encode_sampling_out_types():
    from sentencepiece_amd.processor import SentencePieceProcessor
    check_python_encode_sampling_out_types(lambda blob: SentencePieceProcessor(model_proto=blob))


def check_wrapper_spellings(make_sp):
    """The rest of the reference wrapper's spellings on this path, against the reference module where it is installed."""
    blob = fixtures.model_blob("test_model")
    sp = make_sp(blob)
    s = "Hello world, this is a test."
    assert sp.Tokenize(s) == sp.Encode(s) and sp.Detokenize(sp.Encode(s)) == sp.Decode(sp.Encode(s))
    assert sp.DecodePieces(sp.EncodeAsPieces(s)) == sp.Decode(sp.Encode(s))
    assert sp.serialized_model_proto() == blob and sp.get_piece_size() == sp.GetPieceSize()
    assert sp.NBestEncode(s, out_type=int, nbest_size=3) == sp.NBestEncodeAsIds(s, 3)
    assert sp.NBestEncode(s, out_type=str, nbest_size=3) == sp.NBestEncodeAsPieces(s, 3)
    with_opts = sp.NBestEncode(s, out_type=str, nbest_size=2, add_bos=True, add_eos=True, reverse=True)
    assert [r[0] for r in with_opts] == ["<s>"] * 2 and [r[-1] for r in with_opts] == ["</s>"] * 2
    assert [r[1:-1][::-1] for r in with_opts] == sp.NBestEncodeAsPieces(s, 2)
    imm = sp.NBestEncodeAsImmutableProto(s, 3)
    assert len(imm.nbests) == 3 and imm.SerializeAsString() == sp.NBestEncodeAsSerializedProto(s, 3)
    assert [[p.piece for p in v.pieces] for v in imm.nbests] == sp.NBestEncodeAsPieces(s, 3)
    assert imm.nbests[0].score >= imm.nbests[1].score >= imm.nbests[2].score
    assert sp.SampleEncodeAsIds(s, 1, 0.5) == sp.Encode(s)
    assert sp.SampleEncodeAsImmutableProto(s, -1, 0.2).text == s
    try:
        import sentencepiece as ref_mod
    except ImportError:
        return
    ref = ref_mod.SentencePieceProcessor(model_proto=blob)
    assert [sp.GetScore(i) for i in range(0, sp.GetPieceSize(), 37)] == [ref.GetScore(i) for i in range(0, ref.GetPieceSize(), 37)]
    assert sp.NBestEncode(s, out_type=str, nbest_size=4, add_bos=True, reverse=True) == ref.NBestEncode(s, out_type=str, nbest_size=4, add_bos=True, reverse=True)
    assert sp.NBestEncode([s, "x"], out_type=int, nbest_size=2, add_eos=True) == ref.NBestEncode([s, "x"], out_type=int, nbest_size=2, add_eos=True)
    assert sp.DecodePieces(ref.EncodeAsPieces(s)) == ref.DecodePieces(ref.EncodeAsPieces(s))


def test_emu_wrapper_spellings(emu):
    check_wrapper_spellings(lambda blob: emu.load(blob).sp)


@pytest.mark.gpu
def test_gpu_wrapper_spellings():
    from sentencepiece_amd.processor import SentencePieceProcessor
    check_wrapper_spellings(lambda blob: SentencePieceProcessor(model_proto=blob))
