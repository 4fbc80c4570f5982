"""Seeded synthetic corpora for the BASELINE.json configs.

Bench / test infrastructure: produces the packed sentence buffer the C-ABI
takes -- one ``uint8`` text blob plus ``uint64`` offsets (n + 1) -- without
going through Python string objects, so 10 M sentences can be generated on the
bench host in well under a minute.

C2/C3/C4 generator (``ascii_corpus``): a word list of ``n_words`` synthetic
lower-case words (geometric lengths, mean ~4.7, clipped to [1, 12]) with
Zipf(1.1) frequencies; a sentence is a run of words joined by single spaces
whose target byte length is lognormal, clipped to [8, 512], mean ~128.  A few
"special" entries in the word list supply the rare events SURVEY.md asks for:
empty words (doubled / leading / trailing spaces), punctuation bursts and digit
runs (score ties), and non-ASCII / malformed byte strings (charsmap rules,
U+FFFD fallback).  Sentences are emitted sorted by byte length ("length
bucketed"), as BASELINE.json specifies.

C5 generator (``mixed_corpus``): mixed-script UTF-8 with power-law lengths on
[16, 4096].
"""
from __future__ import annotations

import os

import numpy as np

_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LETTER_P = np.array(
    [12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4,
     2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
_LETTER_P = _LETTER_P / _LETTER_P.sum()

_SPECIAL_WORDS = [
    # (bytes, relative weight inside the "special" probability mass)
    (b"", 6.0),                       # doubled / leading / trailing spaces
    (b"......", 1.0), (b".", 2.0), (b"...", 1.0), (b"!!!", 0.5), (b"?!", 0.5),
    (b"--", 0.5), (b",", 2.0), (b"2024", 0.6), (b"3.14159", 0.3), (b"000000", 0.3),
    (b"1,000,000", 0.3), (b"(a)", 0.3), (b"\t", 0.2),
    ("café".encode(), 0.08), ("naïve".encode(), 0.05),
    ("über".encode(), 0.05), ("日本".encode(), 0.05),
    ("€".encode(), 0.03), ("ＡＢ".encode(), 0.03),   # full-width AB (NFKC rule)
    ("　".encode(), 0.03),                                     # ideographic space
    (b"\xff", 0.02), (b"\xe2\x96", 0.01), (b"A\xcc\x8a", 0.02),   # malformed, truncated, A + ring
]
_SPECIAL_MASS = 0.012   # ~1.2 % of word draws; ~3 % of sentences carry one


class WordList:
    """The fixed synthetic vocabulary a corpus is drawn from."""

    def __init__(self, n_words: int = 50000, seed: int = 20250227, max_word_len: int = 12, mean_word_len: float = 4.7):
        rng = np.random.default_rng(seed)
        lens = np.clip(rng.geometric(1.0 / mean_word_len, size=n_words), 1, max_word_len).astype(np.int64)
        # the most frequent words are short, as in natural text
        lens[:64] = np.clip(lens[:64], 1, 4)
        letters = _LETTERS[rng.choice(len(_LETTERS), size=int(lens.sum()), p=_LETTER_P)]
        specials = [w for w, _ in _SPECIAL_WORDS]
        sp_lens = np.array([len(w) for w in specials], dtype=np.int64)
        self.lens = np.concatenate([lens, sp_lens])
        self.blob = np.concatenate(
            [letters, np.frombuffer(b"".join(specials), dtype=np.uint8)]).astype(np.uint8)
        self.offs = np.concatenate([[0], np.cumsum(self.lens)])[:-1].astype(np.int64)
        zipf = 1.0 / np.arange(1, n_words + 1) ** 1.1
        zipf = zipf / zipf.sum() * (1.0 - _SPECIAL_MASS)
        spw = np.array([w for _, w in _SPECIAL_WORDS])
        spw = spw / spw.sum() * _SPECIAL_MASS
        self.cdf = np.cumsum(np.concatenate([zipf, spw]))
        self.cdf[-1] = 1.0
        self.mean_len = float((np.concatenate([zipf, spw]) * self.lens).sum())


def _target_lengths(rng, n, mean=128.0, lo=8, hi=512, sigma=0.6):
    mu = np.log(mean) - sigma * sigma / 2.0
    return np.clip(rng.lognormal(mu, sigma, size=n), lo, hi)


def ascii_corpus(n_sentences: int, seed: int = 20250227, words: WordList | None = None,
                 mean_len: float = 128.0, sort_by_length: bool = True,
                 chunk: int = 250_000, workers: int | None = None):
    """Returns (text uint8[total], offsets uint64[n + 1]).

    With ``sort_by_length`` the *target* lengths are sorted before the words
    are drawn, so the packed buffer is length-bucketed (neighbouring sentences
    differ by at most one word) without a second pass over the bytes.
    Blocks of ``chunk`` sentences are generated independently (own RNG stream
    ``[seed, block]``) and, for large corpora, in a process pool.
    """
    words = words or WordList()
    target = _target_lengths(np.random.default_rng([seed, 0x7A67]), n_sentences, mean=mean_len)
    if sort_by_length:
        target.sort()
    jobs = [(target[a:a + chunk], seed, b, words)
            for b, a in enumerate(range(0, n_sentences, chunk))]
    if workers is None:
        # (several ranks of one host generate at once: SPMX_SYNTH_WORKERS -- bench.py sets cpu_count // world -- caps the pool)
        cap = int(os.environ.get("SPMX_SYNTH_WORKERS", "32") or 32)
        workers = max(1, min(len(jobs), os.cpu_count() or 1, 32, cap))
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        # close + join, not the context manager: its terminate() SIGTERMs the workers, and under rocprofv3 (whose tool
        # library the forked workers inherit) each one then runs the profiler's signal handler -- a --pmc pass hung there
        pool = mp.get_context("fork").Pool(workers)
        try:
            parts = pool.map(_ascii_job, jobs)
        finally:
            pool.close()
            pool.join()
    else:
        parts = [_ascii_job(j) for j in jobs]
    text = np.concatenate([p[0] for p in parts]) if len(parts) > 1 else parts[0][0]
    lens = np.concatenate([p[1] for p in parts]) if len(parts) > 1 else parts[0][1]
    offs = np.zeros(n_sentences + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    return text, offs


def _ascii_job(job):
    target, seed, block, words = job
    return _ascii_block(target, np.random.default_rng([seed, block]), words)


def _ascii_block(target, rng, words):
    m = len(target)
    # draw enough words to cover the block, then cut sentence boundaries where
    # the running byte count crosses the running target.
    n_draw = int(target.sum() / (words.mean_len + 1.0) * 1.02) + 64
    wid = np.searchsorted(words.cdf, rng.random(n_draw), side="right")
    wl = words.lens[wid]
    run = np.cumsum(wl + 1)                       # bytes incl. one separator each
    cuts = np.searchsorted(run, np.cumsum(target), side="left") + 1   # exclusive word index
    ar = np.arange(m)
    cuts = np.maximum.accumulate(np.maximum(cuts, 1) - ar) + ar   # >= 1 word each, strictly increasing
    n_used = int(cuts[-1])
    if n_used > n_draw:
        raise RuntimeError("synthetic corpus under-drew words")
    wid, wl, run = wid[:n_used], wl[:n_used], run[:n_used]
    # Output layout: every word is followed by one byte; that byte is a space
    # inside a sentence and is dropped (not emitted) after a sentence's last
    # word.  We first lay words + separators out, then delete the separators
    # at sentence ends.
    total = int(run[-1])
    out = np.full(total, 0x20, dtype=np.uint8)
    starts = run - (wl + 1)
    nz = wl > 0
    wl_nz = wl[nz].astype(np.int32)
    rep_src = np.repeat((words.offs[wid[nz]] - starts[nz]).astype(np.int32), wl_nz)
    gaps = np.ones(total, dtype=bool)
    gaps[run - 1] = False                          # separator slots
    pos = np.flatnonzero(gaps).astype(np.int32)    # byte slots of all words, in order
    out[pos] = words.blob[pos + rep_src]
    keep = np.ones(total, dtype=bool)
    keep[run[cuts - 1] - 1] = False               # separator after each last word
    text = out[keep]
    ends = run[cuts - 1] - 1 - np.arange(m)       # sentence end offsets after deletion
    lens = np.diff(np.concatenate([[0], ends]))
    return text, lens.astype(np.uint64)


def _ragged_arange(lens):
    total = int(lens.sum())
    starts = np.cumsum(lens) - lens
    return np.arange(total, dtype=np.int64) - np.repeat(starts, lens)


def sort_packed_by_length(text, offs):
    """Stable sort of a packed buffer by sentence byte length."""
    lens = np.diff(offs.astype(np.int64))
    order = np.argsort(lens, kind="stable")
    return gather_packed(text, offs, order)


def gather_packed(text, offs, order):
    offs_i = offs.astype(np.int64)
    lens = np.diff(offs_i)[order]
    new_offs = np.zeros(len(order) + 1, dtype=np.uint64)
    np.cumsum(lens, out=new_offs[1:])
    src = np.repeat(offs_i[:-1][order] - new_offs[:-1].astype(np.int64), lens)
    idx = np.arange(int(lens.sum()), dtype=np.int64)
    return text[idx + src], new_offs


def pack(sentences):
    """list[bytes | str] -> (text uint8, offsets uint64)."""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in sentences]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=offs[1:])
    text = np.frombuffer(b"".join(bs), dtype=np.uint8).copy()
    return text, offs


def unpack(text, offs):
    b = text.tobytes()
    return [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]



def open_vocab_corpus(n_sentences: int, seed: int = 20250301, oov: float = 0.05, words: WordList | None = None,
                      sort_by_length: bool = True):
    """The C2 recipe with a share `oov` of the word draws replaced by FRESH random words (letters by frequency, length
    geometric with mean 6, clipped to [2, 14]) that are in no word list and in no model trained on one: text the
    load-time word memo does not fit by construction.  -> (text uint8, offsets uint64)."""
    words = words or WordList()
    rng = np.random.default_rng([seed, 0x00F])
    n_fresh = 200_000
    lens = np.clip(rng.geometric(1.0 / 6.0, size=n_fresh), 2, 14).astype(np.int64)
    letters = _LETTERS[rng.choice(len(_LETTERS), size=int(lens.sum()), p=_LETTER_P)]
    w2 = WordList.__new__(WordList)
    w2.lens = np.concatenate([words.lens, lens])
    w2.blob = np.concatenate([words.blob, letters]).astype(np.uint8)
    w2.offs = np.concatenate([[0], np.cumsum(w2.lens)])[:-1].astype(np.int64)
    p_old = np.diff(np.concatenate([[0.0], words.cdf])) * (1.0 - oov)
    p_new = np.full(n_fresh, oov / n_fresh)
    w2.cdf = np.cumsum(np.concatenate([p_old, p_new]))
    w2.cdf[-1] = 1.0
    w2.mean_len = float((np.concatenate([p_old, p_new]) * w2.lens).sum())
    return ascii_corpus(n_sentences, seed=seed, words=w2, sort_by_length=sort_by_length)


def repeated_file_corpus(path: str, times: int, seed: int = 20250302):
    """A text file's lines, `times` over, every repetition with its lines in another order: natural text at bench size
    (data/botchan.txt x 2000 = 8.6 M lines).  -> (text uint8, offsets uint64)."""
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    t0, o0 = pack(lines)
    n = len(lines)
    rng = np.random.default_rng(seed)
    lens0 = np.diff(o0.astype(np.int64))
    order = np.concatenate([rng.permutation(n) for _ in range(times)])
    lens = lens0[order]
    offs = np.zeros(len(order) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    src = np.repeat(o0.astype(np.int64)[:-1][order] - offs[:-1].astype(np.int64), lens)
    text = t0[np.arange(int(lens.sum()), dtype=np.int64) + src]
    return text, offs

# ---------------------------------------------------------------- config 5 --
_CJK_LO, _CJK_HI = 0x4E00, 0x9FA5


def mixed_corpus(n_sentences: int, seed: int = 20250228, lo: int = 16, hi: int = 4096,
                 alpha: float = 1.5, sort_by_length: bool = True):
    """Mixed-script UTF-8, 60 % CJK / 30 % Latin / 10 % other, power-law lengths."""
    rng = np.random.default_rng(seed)
    u = rng.random(n_sentences)
    a = 1.0 - alpha
    target = ((hi ** a - lo ** a) * u + lo ** a) ** (1.0 / a)
    n_chars = int(target.sum() / 2.2) + n_sentences * 4
    kind = rng.random(n_chars)
    cp = np.empty(n_chars, dtype=np.int64)
    cjk = kind < 0.60
    lat = (kind >= 0.60) & (kind < 0.90)
    oth = kind >= 0.90
    # Zipf-ish CJK: a small hot set plus a long tail
    hot = rng.zipf(1.3, size=int(cjk.sum())) % (_CJK_HI - _CJK_LO)
    cp[cjk] = _CJK_LO + hot
    lat_pool = np.frombuffer(b"etaoinshrdlcumwfgypbvk      ", dtype=np.uint8)
    cp[lat] = lat_pool[rng.integers(0, len(lat_pool), size=int(lat.sum()))]
    oth_pool = np.concatenate([
        np.arange(0x3041, 0x3097), np.arange(0x30A1, 0x30FB),      # kana (3 B)
        np.arange(0x0410, 0x0450), np.arange(0x00C0, 0x0100),      # cyrillic, latin-1 (2 B)
        np.arange(0xFF21, 0xFF3B), np.array([0x3000, 0x2460, 0x337F]),   # NFKC rules
        np.arange(0x1F600, 0x1F640),                               # emoji (4 B)
    ])
    cp[oth] = oth_pool[rng.integers(0, len(oth_pool), size=int(oth.sum()))]
    nb = np.where(cp < 0x80, 1, np.where(cp < 0x800, 2, np.where(cp < 0x10000, 3, 4)))
    run = np.cumsum(nb)
    cuts = np.searchsorted(run, np.cumsum(target), side="left") + 1
    ar = np.arange(n_sentences)
    cuts = np.maximum.accumulate(np.maximum(cuts, 1) - ar) + ar
    if int(cuts[-1]) > n_chars:
        raise RuntimeError("synthetic corpus under-drew characters")
    n_used = int(cuts[-1])
    cp, nb, run = cp[:n_used], nb[:n_used], run[:n_used]
    total = int(run[-1])
    out = np.zeros(total, dtype=np.uint8)
    st = run - nb
    for k in (1, 2, 3, 4):
        m = nb == k
        c, s = cp[m], st[m]
        if k == 1:
            out[s] = c
        elif k == 2:
            out[s] = 0xC0 | (c >> 6); out[s + 1] = 0x80 | (c & 0x3F)
        elif k == 3:
            out[s] = 0xE0 | (c >> 12); out[s + 1] = 0x80 | ((c >> 6) & 0x3F)
            out[s + 2] = 0x80 | (c & 0x3F)
        else:
            out[s] = 0xF0 | (c >> 18); out[s + 1] = 0x80 | ((c >> 12) & 0x3F)
            out[s + 2] = 0x80 | ((c >> 6) & 0x3F); out[s + 3] = 0x80 | (c & 0x3F)
    offs = np.zeros(n_sentences + 1, dtype=np.uint64)
    offs[1:] = run[cuts - 1]
    if sort_by_length:
        return sort_packed_by_length(out, offs)
    return out, offs


# ------------------------------------------------- config 5: 250k vocabulary --
def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _top_level_fields(blob: bytes):
    """proto2 wire format: yields (field number, wire type, payload bytes incl. nothing of the key)."""
    i, n = 0, len(blob)
    while i < n:
        key = 0
        shift = 0
        while True:
            b = blob[i]; i += 1
            key |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        field, wt = key >> 3, key & 7
        if wt == 2:
            ln = 0
            shift = 0
            while True:
                b = blob[i]; i += 1
                ln |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            yield field, wt, blob[i:i + ln]
            i += ln
        elif wt == 0:
            j = i
            while blob[j] & 0x80:
                j += 1
            yield field, wt, blob[i:j + 1]
            i = j + 1
        elif wt == 5:
            yield field, wt, blob[i:i + 4]
            i += 4
        elif wt == 1:
            yield field, wt, blob[i:i + 8]
            i += 8
        else:
            raise ValueError("unsupported wire type %d" % wt)


def _piece_msg(piece: bytes, score: float, ptype: int) -> bytes:
    # SentencePiece { 1: piece, 2: score (float), 3: type } (src/sentencepiece_model.proto:293-310)
    body = b"\x0a" + _varint(len(piece)) + piece + b"\x15" + np.float32(score).tobytes()
    if ptype != 1:
        body += b"\x18" + _varint(ptype)
    return b"\x0a" + _varint(len(body)) + body


def requantized_model(model: bytes, quantum: float = 0.5) -> bytes:
    """``model`` with every piece score rounded to a multiple of ``quantum``: equal-score candidates (and equal rounded
    sums of different paths) become common -- the tie handling of the Viterbi folds is what such a model tests."""
    return rescored_model(model, lambda v: round(v / quantum) * quantum)


def rescored_model(model: bytes, fn) -> bytes:
    """``model`` with every piece score v replaced by fn(v)."""
    out = bytearray()
    for field, wt, payload in _top_level_fields(model):
        if field == 1 and wt == 2:
            piece, score, ptype = b"", 0.0, 1
            for f2, w2, p2 in _top_level_fields(payload):
                if f2 == 1:
                    piece = p2
                elif f2 == 2:
                    score = float(np.frombuffer(p2, dtype=np.float32)[0])
                elif f2 == 3:
                    ptype = int(p2[0])
            out += _piece_msg(piece, fn(score), ptype)
        elif wt == 2:
            out += _varint(field << 3 | 2) + _varint(len(payload)) + payload
        else:
            out += _varint(field << 3 | wt) + payload
    return bytes(out)


def c5_model(template_model: bytes, vocab_size: int = 250_000, seed: int = 20250228,
             byte_fallback: bool = False, sample_sentences: int = 30_000) -> bytes:
    """A ``vocab_size``-piece unigram ModelProto for BASELINE.json configs[4].

    No training corpus of that size exists here, so the vocabulary is synthesized: every character of a
    ``mixed_corpus`` sample, then random 2-6 character substrings of it (spaces written as U+2581), most
    frequent first, with log-rank scores partly quantized so that equal-score candidates occur.  trainer_spec
    and normalizer_spec (nmt_nfkc charsmap) are taken verbatim from ``template_model`` (a serialized 32k
    model); ``byte_fallback`` appends the 256 byte pieces and sets trainer_spec.byte_fallback.
    Deterministic for a given numpy version."""
    rng = np.random.default_rng(seed)
    text, offs = mixed_corpus(sample_sentences, seed=seed + 1, sort_by_length=False)
    raw = text.tobytes()
    s = raw.decode("utf-8").replace(" ", "▁")
    chars = np.array(list(s))
    # single characters by frequency
    uniq, cnt = np.unique(chars, return_counts=True)
    order = np.argsort(-cnt, kind="stable")
    pieces = ["<unk>", "<s>", "</s>"]
    types = [2, 3, 3]
    scores = [0.0, 0.0, 0.0]
    if byte_fallback:
        for b in range(256):
            pieces.append("<0x%02X>" % b); types.append(6); scores.append(0.0)
    seen = set(pieces)
    singles = [str(uniq[i]) for i in order]
    # substrings: sample (start, length) pairs, count, keep the most frequent
    n_draw = vocab_size * 12
    st = rng.integers(0, len(chars) - 8, size=n_draw)
    ln = rng.integers(2, 7, size=n_draw)
    from collections import Counter
    c = Counter("".join(chars[a:a + l]) for a, l in zip(st.tolist(), ln.tolist()))
    # a piece never has U+2581 after its first character (as with split_by_whitespace=true)
    subs = [w for w, _ in c.most_common() if "▁" not in w[1:]]
    body = []
    for w in singles:
        if w not in seen:
            seen.add(w); body.append(w)
    n_single = len(body)
    for w in subs:
        if len(pieces) + len(body) >= vocab_size:
            break
        if w not in seen:
            seen.add(w); body.append(w)
    rank = np.arange(1, len(body) + 1, dtype=np.float64)
    sc = -(2.5 + 1.15 * np.log(rank)) - 0.35 * np.array([len(w) for w in body])
    sc[:n_single] -= 1.5                       # single characters are the fallback, not the preferred split
    sc += rng.normal(0.0, 0.2, size=len(body))
    q = rng.random(len(body)) < 0.5
    sc[q] = np.round(sc[q] * 4.0) / 4.0        # exact ties between candidates
    out = bytearray()
    for p, t, v in zip(pieces, types, scores):
        out += _piece_msg(p.encode("utf-8"), v, t)
    for w, v in zip(body, sc.tolist()):
        out += _piece_msg(w.encode("utf-8"), v, 1)
    for field, wt, payload in _top_level_fields(template_model):
        if field == 2:      # trainer_spec
            if byte_fallback:
                payload = payload + b"\x98\x02\x01"      # field 35 (byte_fallback) = true, last one wins
            out += b"\x12" + _varint(len(payload)) + payload
        elif field == 3:    # normalizer_spec
            out += b"\x1a" + _varint(len(payload)) + payload
    return bytes(out)
