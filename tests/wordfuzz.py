"""Input generators aimed at the WORD form of the device encoder (csrc/kernels_word.h): the kernels that take a
sentence a word at a time through the load-time and call-local word memos.  Shared by tests/test_word_form.py (CPU
emulator and -m gpu legs) and scripts/fuzz_wordmemo.py (the open-ended campaign).  TEST INFRASTRUCTURE ONLY.

  * nul_sentences / control_corpus: vocabulary words with U+0000, the other control bytes, 0x7F and stray UTF-8 lead /
    continuation bytes spliced in front of, inside and behind them.  The reference keeps U+0000 as a character of its
    own (src/normalizer.cc:231-244) and gives it <unk> (src/unigram_model.cc:995-1005); a memo keyed by zero-padded
    bytes lost it (round-3 verdict).
  * near_tie_model / near_tie_corpus: random unigram models whose scores are QUANTIZED (exact ties between
    segmentations) or a few float ulps apart (decisions that flip with the magnitude of the accumulated score,
    src/unigram_model.cc:979-989) and long sentences of the models' own words -- what the memo's margin guard is for.
"""
import numpy as np

from sentencepiece_amd import synth

SP = "▁"

# every byte that is not a plain word byte: C0 controls with NUL first, space, DEL, lone continuation / lead bytes,
# bytes that are never valid in UTF-8
CONTROL_BYTES = [bytes([b]) for b in range(0x00, 0x21)] + [b"\x7f", b"\x80", b"\xbf", b"\xc3", b"\xe3", b"\xf0", b"\xff", b"\xc0"]


def whole_words(model_blob, limit=4000):
    """The model's vocabulary strings that are whole plain words (U+2581 + 1..16 bytes 0x21..0x7E): the memo's keys."""
    from sentencepiece import sentencepiece_model_pb2 as pb   # (only to list piece strings)
    m = pb.ModelProto()
    m.ParseFromString(model_blob)
    out = []
    for p in m.pieces:
        s = p.piece
        if p.type == 1 and s.startswith(SP) and 1 < len(s) <= 17:
            body = s[1:]
            if all(0x21 <= ord(c) <= 0x7E for c in body):
                out.append(body.encode())
    out.sort(key=lambda w: (len(w), w))
    if len(out) > limit:
        step = len(out) / limit
        out = [out[int(i * step)] for i in range(limit)]
    return out


def nul_sentences(words):
    """The round-3 repro, spelled out: a memo word followed by one / two NULs at the sentence's start, middle, end."""
    w = [x for x in words if 2 <= len(x) <= 6][:40] + [x for x in words if len(x) in (11, 12, 13, 15, 16)][:24]
    sents = []
    for i, a in enumerate(w):
        b, c = w[(i + 1) % len(w)], w[(i + 7) % len(w)]
        for z in (b"\x00", b"\x00\x00"):
            sents += [a + z, a + z + b" " + b, b + b" " + a + z, b + b" " + a + z + b" " + c, z + a + b" " + b,
                      a + b" " + z + b" " + b, a + z + b, a[:1] + z + a[1:] + b" " + b, a + z + b"  " + b + z]
    return sents


def control_corpus(words, n, seed):
    """n sentences of vocabulary words with CONTROL_BYTES spliced in.  Most sentences carry one or two splices, so that
    the rest of the sentence stays in the word form and the splice is what decides."""
    rng = np.random.default_rng(seed)
    nw = len(words)
    sents = []
    for _ in range(n):
        k = int(rng.choice([1, 2, 4, 8, 20, 40]))
        ws = [bytearray(words[int(i)]) for i in rng.integers(0, nw, size=k)]
        for _ in range(int(rng.choice([0, 1, 1, 1, 2, 3]))):
            j = int(rng.integers(0, k))
            cb = CONTROL_BYTES[int(rng.integers(0, len(CONTROL_BYTES)))] * int(rng.choice([1, 1, 1, 2, 3]))
            how = int(rng.integers(0, 5))
            if how == 0:
                ws[j] = ws[j] + cb                      # behind the word (the round-3 failure when cb is NUL)
            elif how == 1:
                ws[j] = bytearray(cb) + ws[j]           # in front of it
            elif how == 2:
                at = int(rng.integers(0, len(ws[j]) + 1))
                ws[j] = ws[j][:at] + cb + ws[j][at:]    # inside
            elif how == 3:
                ws[j] = bytearray(cb)                   # a token of its own
            else:
                ws[j] = ws[j] + cb + ws[(j + 1) % k]    # two words glued by it
        s = b" ".join(bytes(x) for x in ws)
        r = rng.random()
        if r < 0.05:
            s = b" " + s
        elif r < 0.10:
            s = s + b" "
        elif r < 0.15:
            s = s.replace(b" ", b"  ", 1)
        sents.append(s)
    return sents


def near_tie_model(rng, base_blob):
    """-> (serialized ModelProto, its words).  See the module docstring."""
    from sentencepiece import sentencepiece_model_pb2 as pb
    m = pb.ModelProto()
    m.ParseFromString(base_blob)
    del m.pieces[:]
    for name, typ in (("<unk>", 2), ("<s>", 3), ("</s>", 3)):
        p = m.pieces.add()
        p.piece, p.score, p.type = name, 0.0, typ
    alpha = "abcde"[:int(rng.integers(2, 6))] + ("." if rng.random() < 0.5 else "")
    quant = float(rng.choice([1.0, 0.5, 0.125, 2.0 ** -10, 0.0]))
    tiny = float(rng.choice([0.0, 2.0 ** -20, 2.0 ** -16, 2.0 ** -12, 1e-3]))

    def score():
        v = -float(rng.uniform(1.0, 14.0))
        if quant:
            v = round(v / quant) * quant
        if tiny and rng.random() < 0.5:
            v += tiny * float(rng.integers(-3, 4))
        return float(np.float32(v))
    seen = {}

    def add(s, v=None):
        if s and s not in seen:
            seen[s] = float(np.float32(score() if v is None else v))
            p = m.pieces.add()
            p.piece, p.score, p.type = s, seen[s], 1
    add(SP)
    for c in alpha:
        if rng.random() < 0.9:
            add(c)
        if rng.random() < 0.7:
            add(SP + c)
    words = []
    for _ in range(int(rng.integers(60, 300))):
        w = "".join(alpha[int(k)] for k in rng.integers(0, len(alpha), size=int(rng.integers(2, 9))))
        words.append(w)
        # a split of the word into two pieces, and the whole word scored a hair above / below / exactly at the split's sum
        k = int(rng.integers(1, len(w)))
        a, b = SP + w[:k], w[k:]
        add(a)
        add(b)
        if rng.random() < 0.97:
            delta = float(rng.choice([0.0, 2.0 ** -22, -2.0 ** -22, 2.0 ** -18, -2.0 ** -18, 1e-4, -1e-4, 1e-2, -1e-2, 1.0, -1.0]))
            add(SP + w, np.float32(seen[a]) + np.float32(seen[b]) + np.float32(delta))
    return m.SerializeToString(), words


def near_tie_corpus(rng, words, n, p_len=None):
    """p_len: probabilities of the sentence lengths 1, 3, 10, 40, 150, 400 words (default: uniform)."""
    sents = []
    for _ in range(n):
        k = int(rng.choice([1, 3, 10, 40, 150, 400], p=p_len))
        ws = [words[int(i)] for i in rng.integers(0, len(words), size=k)]
        s = " ".join(ws)
        if rng.random() < 0.1:
            s = "  " + s + " "
        if rng.random() < 0.05:
            s = s.replace(" ", "  ", 1)
        sents.append(s.encode())
    return synth.pack(sents)


def first_difference(ids, io, oids, oio):
    """Index of the first sentence whose ids differ (-1: none) -- for assertion messages."""
    if np.array_equal(io, oio) and np.array_equal(ids, oids):
        return -1
    a, b = np.asarray(io).astype(np.int64), np.asarray(oio).astype(np.int64)
    for s in range(min(len(a), len(b)) - 1):
        if ids[a[s]:a[s + 1]].tolist() != oids[b[s]:b[s + 1]].tolist():
            return s
    return min(len(a), len(b)) - 1
