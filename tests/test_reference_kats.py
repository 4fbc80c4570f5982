"""Known-answer tests transcribed from the reference's own unit tests: the tiny hand-built
models of src/unigram_model_test.cc:782-871 (EncodeTest), :873-928 (EncodeWithUnusedTest) and
src/bpe_model_test.cc:49-141 (EncodeTest), :143-187 (EncodeAmbiguousTest), :195-250 (EncodeWithUnusedTest), rebuilt
as ModelProtos with an identity normalizer, and their expected piece sequences turned into ids the way
PopulateSentencePieceText does (PieceToId; a run of unknown pieces is one unk id,
src/sentencepiece_processor.cc:609-613).

Checked: the oracle, the compiled reference where it is built (this pins the transcription itself), the device
kernels under the emulator, and -m gpu the HIP path."""
import numpy as np
import pytest

from sentencepiece_amd import synth
from tests import refshim

UNK, CONTROL, USER_DEFINED, UNUSED = 2, 3, 4, 5


def build_model(model_type, pieces):
    """pieces: [(string, score, type)] after <unk>, <s>, </s> (MakeBaseModelProto)."""
    out = bytearray()
    for p, s, t in [("<unk>", 0.0, UNK), ("<s>", 0.0, CONTROL), ("</s>", 0.0, CONTROL)] + pieces:
        out += synth._piece_msg(p.encode("utf-8"), s, t)
    trainer = b"\x18" + synth._varint(model_type)                       # trainer_spec.model_type = 3
    out += b"\x12" + synth._varint(len(trainer)) + trainer
    norm = b"\x0a\x08identity" + b"\x18\x00" + b"\x20\x00" + b"\x28\x01"   # name, add_dummy_prefix=0, remove_extra_ws=0, escape=1
    out += b"\x1a" + synth._varint(len(norm)) + norm
    return bytes(out)


ENCODE_PIECES = [("ab", 0.0, 1), ("cd", -0.1, 1), ("abc", -0.2, 1), ("a", -0.3, 1), ("b", -0.4, 1), ("c", -0.5, 1),
                 ("ABC", -0.5, USER_DEFINED), ("abcdabcd", -0.5, USER_DEFINED), ("q", -0.5, USER_DEFINED),
                 ("r", -0.5, USER_DEFINED), ("qr", -0.5, 1)]
ENCODE_KATS = [("", []), ("abc", ["abc"]), ("AB", ["A", "B"]), ("abcd", ["ab", "cd"]), ("abcc", ["abc", "c"]),
               ("xabcabaabcdd", ["x", "abc", "ab", "a", "ab", "cd", "d"]),
               ("xyz東京", ["x", "y", "z", "東", "京"]), ("ABC", ["ABC"]), ("abABCcd", ["ab", "ABC", "cd"]),
               ("ababcdabcdcd", ["ab", "abcdabcd", "cd"]), ("abqrcd", ["ab", "q", "r", "cd"])]
AMBIG_PIECES = [("aa", -0.1, 1), ("bb", -0.2, 1), ("ab", -0.3, 1), ("a", -0.4, 1), ("b", -0.5, 1)]
AMBIG_KATS = [("aaa", ["aa", "a"]), ("aabb", ["aa", "bb"]), ("aaabbb", ["aa", "a", "bb", "b"]),
              ("aaaba", ["aa", "ab", "a"]), ("あ".encode()[:1], ["\x00broken"])]   # a broken UTF-8 byte: one unknown piece


def unused_pieces(unused):
    ps = [("abcd", 10.0), ("abc", 5.0), ("ab", 2.0), ("cd", 1.0), ("a", 0.0), ("b", 0.0), ("c", 0.0), ("d", 0.0)]
    return [(p, s, UNUSED if (3 + i) in unused else 1) for i, (p, s) in enumerate(ps)]


# (name, model type 1 unigram / 2 bpe, pieces, [(input, expected pieces)])
CASES = [
    ("unigram_encode", 1, ENCODE_PIECES, ENCODE_KATS),
    ("bpe_encode", 2, ENCODE_PIECES, ENCODE_KATS),
    ("bpe_ambiguous", 2, AMBIG_PIECES, AMBIG_KATS),
    ("unigram_unused_none", 1, unused_pieces(()), [("abcd", ["abcd"])]),
    ("unigram_unused_3", 1, unused_pieces((3,)), [("abcd", ["abc", "d"])]),
    ("unigram_unused_3_5", 1, unused_pieces((3, 5)), [("abcd", ["abc", "d"])]),
    ("unigram_unused_3_4", 1, unused_pieces((3, 4)), [("abcd", ["ab", "cd"])]),
    ("bpe_unused_none", 2, unused_pieces(()), [("abcd", ["abcd"])]),
    ("bpe_unused_3", 2, unused_pieces((3,)), [("abcd", ["abc", "d"])]),
    ("bpe_unused_3_5", 2, unused_pieces((3, 5)), [("abcd", ["abc", "d"])]),
    ("bpe_unused_3_4", 2, unused_pieces((3, 4)), [("abcd", ["ab", "c", "d"])]),
]


def expected_ids(pieces, kats):
    ids_of = {p: 3 + i for i, (p, _, _) in enumerate(pieces)}
    out = []
    for _, exp in kats:
        ids = []
        for w in exp:
            i = ids_of.get(w, 0)
            if i == 0 and ids and ids[-1] == 0:
                continue                                   # a run of unknown pieces yields one id
            ids.append(i)
        out.append(ids)
    return out


def packed(kats):
    return synth.pack([k if isinstance(k, bytes) else k.encode("utf-8") for k, _ in kats])


def check(encode_batch, kats, want):
    text, offs = packed(kats)
    ids, io = encode_batch(text, offs)
    io = io.astype(np.int64)
    got = [ids[io[i]:io[i + 1]].tolist() for i in range(len(kats))]
    assert got == want


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_kats_oracle_reference_emulator(case, oracle):
    name, mtype, pieces, kats = case
    blob = build_model(mtype, pieces)
    want = expected_ids(pieces, kats)
    check(oracle.load(blob).encode_batch, kats, want)
    if refshim.available():
        check(lambda t, o: refshim.RefLib().load(blob).encode_batch(t, o, threads=1), kats, want)
    from tests import emulib
    check(lambda t, o: emulib.EmuLib().load(blob).encode_batch(t, o, grid=2), kats, want)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_kats_gpu(case):
    from sentencepiece_amd.processor import SentencePieceProcessor
    name, mtype, pieces, kats = case
    sp = SentencePieceProcessor(model_proto=build_model(mtype, pieces))
    check(sp.EncodePacked, kats, expected_ids(pieces, kats))
