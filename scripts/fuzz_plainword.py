# Structure campaign for the word form (csrc/kernels_word.h) and the plain scan in front of it (csrc/kernels.h
# plain_scan_block): batches of mostly plain-ASCII sentences whose SHAPE is what varies -- word lengths around the memo
# tiers' limits (10 / 11 bytes: two-piece 16-byte entries; 16 / 17: the memo's longest key and the 16-byte window), runs
# of spaces, sentences of spaces only, empty sentences, sentence lengths on the class limits, few fresh words repeated
# often (call-local memo: collect -> resolve -> again), many fresh words once (a full call-local table), sentences
# that are not plain mixed in (the scan's routing and the general launch beside the word round), small call-local
# tables and list capacities.  The device kernels under the emulator against the compiled reference (the oracle where
# it is not built), ids compared sentence by sentence.  usage: python scripts/fuzz_plainword.py SECONDS FIRST_SEED
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sentencepiece_amd import synth
from tests import fixtures, oraclelib, emulib, wordfuzz

MODELS = ["uni32k", "uni32k_w16", "bpe32k", "uni1k", "bpe1k", "test_model", "uni1k_bf", "bpe1k_llama",
          "uni32k+keep_ws", "bpe32k+keep_ws", "uni1k_bf+keep_ws"]     # (+keep_ws: remove_extra_whitespaces switched off)
if os.environ.get("FUZZ_MODELS"):                       # (a campaign over other models: FUZZ_MODELS=c5_250k,c5_250k_bf)
    MODELS = os.environ["FUZZ_MODELS"].split(",")
EDGE_LEN = [0, 1, 2, 3, 15, 16, 17, 19, 20, 21, 23, 24, 25, 31, 32, 33, 39, 40, 41, 63, 64, 65, 127, 128, 191, 192, 193,
            447, 448, 575, 576, 577, 1279, 1535, 1536, 1537, 3328, 4095, 4096, 4097, 6000]
WORD_LEN = [1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 11, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 20, 24, 31, 32, 33, 40, 70]
ALPHA = b"abcdefghijklmnopqrstuvwxyzetaoinshr"
OTHER = b"ABCXYZ0123456789.,;:!?'\"()-_/\\@#$%^&*+=<>[]{}|~`"
NOT_PLAIN = ["\t", "\x00", "\x7f", "é", "ß", "日本", "　", "Ａ", "​", "ﬁ", "\r", "\x1f", "\xa0", "𝒳"]


def fresh_word(rng, vocab):
    """A word the load-time memo does not hold.  Random letters take many pieces (a call-local entry holds few: long
    ones leave the word form), so most are vocabulary words with a letter changed, an affix, or two of them glued."""
    r = rng.random()
    if r < 0.35:
        n = int(rng.choice(WORD_LEN))
        src = ALPHA if rng.random() < 0.8 else ALPHA + OTHER
        return bytes(src[int(i)] for i in rng.integers(0, len(src), size=n))
    w = bytearray(vocab[int(rng.integers(0, len(vocab)))])
    if r < 0.55:
        w[int(rng.integers(0, len(w)))] = ALPHA[int(rng.integers(0, len(ALPHA)))]
    elif r < 0.75:
        w += [b"s", b"ed", b"ing", b"ly", b"'s", b".", b",", b"er", b"ness"][int(rng.integers(0, 9))]
    elif r < 0.85:
        w = bytearray([b"un", b"re", b"(", b"\"", b"x"][int(rng.integers(0, 5))]) + w
    else:
        w += vocab[int(rng.integers(0, len(vocab)))]
    return bytes(w)


def sentence(rng, vocab, pool, target):
    """A plain-ASCII sentence of exactly `target` bytes (when target is not None)."""
    style = rng.random()
    long_ok = rng.random() < 0.25          # words beyond the memo's 16 bytes leave the word form: in a quarter of the sentences
    multi = rng.random() < 0.4             # runs of spaces (a model that keeps them leaves the word form there): in 40 %
    out = bytearray()
    if multi and rng.random() < 0.2:
        out += b" " * int(rng.integers(1, 4))
    limit = target if target is not None else int(rng.choice([5, 20, 60, 130, 300, 900]))
    while len(out) < limit:
        r = rng.random()
        if style < 0.15:
            w = pool[int(rng.integers(0, len(pool)))]                 # a few fresh words, over and over
        elif r < 0.55:
            w = vocab[int(rng.integers(0, len(vocab)))]
        elif r < 0.70:
            w = pool[int(min(len(pool) - 1, rng.zipf(1.3) - 1))]
        elif r < 0.80:
            w = fresh_word(rng, vocab)
        elif r < 0.86:
            w = vocab[int(rng.integers(0, len(vocab)))] + vocab[int(rng.integers(0, len(vocab)))]   # glued words
        elif r < 0.90:
            w = vocab[int(rng.integers(0, len(vocab)))] + bytes([OTHER[int(rng.integers(0, len(OTHER)))]])
        elif r < 0.94:
            w = vocab[int(rng.integers(0, len(vocab)))].upper()
        elif r < 0.97:
            w = bytes([OTHER[int(rng.integers(0, len(OTHER)))]]) * int(rng.integers(1, 5))
        elif multi:
            w = b""                                                   # (a run of spaces)
        else:
            w = b"a"
        if not long_ok and len(w) > 16:
            w = w[:int(rng.choice([9, 10, 11, 15, 16]))]
        out += w
        out += b" " * (int(rng.choice([1, 1, 1, 1, 1, 1, 2, 3, 7])) if multi else 1)
    if target is None:
        if not multi or rng.random() < 0.6:
            while out and out[-1] == 0x20:
                out.pop()
        return bytes(out)
    out = out[:target]
    if out and (not multi or rng.random() < 0.6) and out[-1] == 0x20:
        out[-1] = ord("x")
    return bytes(out)


def batch(rng, vocab):
    pool = [fresh_word(rng, vocab) for _ in range(int(rng.choice([3, 20, 200])))]
    n = int(rng.choice([70, 150, 260]))
    p_other = float(rng.choice([0.0, 0.0, 0.03, 0.15, 0.6]))
    sents = []
    for _ in range(n):
        r = rng.random()
        if r < 0.30:
            s = sentence(rng, vocab, pool, int(rng.choice(EDGE_LEN[:32])))
        elif r < 0.34:
            s = sentence(rng, vocab, pool, int(rng.choice(EDGE_LEN)))
        elif r < 0.37:
            s = b" " * int(rng.choice([1, 2, 3, 16, 17, 40]))
        else:
            s = sentence(rng, vocab, pool, None)
        if rng.random() < p_other and s:
            x = rng.choice(NOT_PLAIN).encode("utf-8", "surrogatepass") if True else b""
            at = int(rng.integers(0, len(s) + 1))
            s = s[:at] + x + s[at:]
        sents.append(s)
    # the batch's last bytes: the plain scan keeps the sentences that end within 20 bytes of the batch's end out of the
    # word form -- make that boundary land everywhere
    for _ in range(int(rng.integers(0, 6))):
        sents.append(sentence(rng, vocab, pool, int(rng.integers(0, 24))))
    return sents


def blob_of(name):
    blob = fixtures.model_blob(name.split("+")[0])
    if name.endswith("+keep_ws"):
        from sentencepiece import sentencepiece_model_pb2 as pb
        m = pb.ModelProto()
        m.ParseFromString(blob)
        m.normalizer_spec.remove_extra_whitespaces = False
        blob = m.SerializeToString()
    return blob


def variant(rng):
    env = {}
    r = rng.random()
    if r < 0.2:
        env["SPMX_NO_WORD_DYN"] = "1"
    elif r < 0.5:
        env["SPMX_DYN_SLOTS_LOG2"] = str(int(rng.choice([4, 5, 7, 10])))     # a call-local table that fills up
        env["SPMX_DYN_LIST_CAP"] = str(int(rng.choice([1, 3, 17, 1000])))
    if rng.random() < 0.25:
        env["SPMX_NO_IDS16"] = "1"
    if rng.random() < 0.15:
        env["SPMX_NO_SCAN"] = "1"
    if rng.random() < 0.15:
        env["SPMX_NO_OVERLAP"] = "1"
    if rng.random() < 0.3:
        env["SPMX_ARENA_FIRST"] = str(int(rng.choice([64, 1000, 20000])))    # the arena grows and the call repeats
    if rng.random() < 0.3:
        env["SPMX_TILE_MIN_LANES"] = "1"
    classes = emulib.SMALL_CLASSES if rng.random() < 0.5 else None
    return env, classes, int(rng.choice([1, 2, 3]))


def main():
    t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 600)
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
    em = emulib.EmuLib()
    try:
        from tests import refshim
        ref = refshim.RefLib() if refshim.available() else None
    except Exception:
        ref = None
    orc = oraclelib.OracleLib()
    blobs = {m: blob_of(m) for m in MODELS}
    checker = {m: (ref or orc).load(blobs[m]) for m in MODELS}
    vocab = {m: wordfuzz.whole_words(blobs[m]) for m in MODELS}
    bad = n_all = n_word = n_batches = 0
    per = {m: [0, 0] for m in MODELS}
    while time.time() < t_end:
        seed += 1
        rng = np.random.default_rng(seed)
        m = MODELS[int(rng.integers(0, len(MODELS)))]
        env, classes, cus = variant(rng)
        sents = batch(rng, vocab[m] or [b"a", b"the"])
        text, offs = synth.pack(sents)
        shift = int(rng.integers(0, 16))                                 # the packed text at any alignment
        raw = np.zeros(len(text) + 64, dtype=np.uint8)
        base = ((-raw.ctypes.data) & 15) + shift
        buf = raw[base:base + len(text)]
        buf[:] = text
        try:
            h = em.load(blobs[m], cus=cus, classes=classes, env=env)
            ids, io = h.encode_batch(buf, offs)
            if ref is not None:
                oi, oo = checker[m].encode_batch(text, offs, threads=2)
            else:
                oi, oo = checker[m].encode_batch(text, offs)
            prof = h.sp.LastProfile()
        except Exception as e:
            bad += 1
            print("EXC seed", seed, m, env, repr(e)[:200], flush=True)
            continue
        n_batches += 1
        n_all += len(sents)
        w = sum(c["sentences"] for c in prof["classes"] if c["kernel"].startswith("EncodeWord"))
        n_word += w
        per[m][0] += w
        per[m][1] += len(sents)
        k = wordfuzz.first_difference(np.asarray(ids), np.asarray(io), np.asarray(oi), np.asarray(oo))
        if h.status or k >= 0:
            bad += 1
            print("MISMATCH seed", seed, m, env, "classes", "small" if classes else "default", "cus", cus, "sentence", k,
                  repr(sents[k][:100]) if k >= 0 else "", "status", h.status, flush=True)
        if n_batches % 50 == 0:
            print("seed", seed, "batches", n_batches, "bad", bad, "sentences", n_all,
                  "word-form share %.3f" % (n_word / max(1, n_all)),
                  " ".join("%s %.2f" % (k, v[0] / max(1, v[1])) for k, v in per.items()), flush=True)
    print("DONE batches", n_batches, "bad", bad, "sentences", n_all, "through the word form", n_word)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
