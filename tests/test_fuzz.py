"""Seeded random inputs beyond the fixtures: raw byte noise (malformed UTF-8), random code points from every
plane, whitespace storms, rule characters between ASCII words, shuffled fragments of the corpora.  Three legs on the same inputs:
  * the oracle against the compiled reference (where it is built: this container) -- pins the oracle on them;
  * the device kernels under the emulator against the oracle (a few hundred sentences per model);
  * -m gpu: the HIP path against the oracle (20 k sentences per model), encode and decode."""
import numpy as np
import pytest

from sentencepiece_amd import synth
from tests import fixtures, refshim

MODELS = ["test_model", "test_ja_model", "uni1k_bf", "uni1k_uds", "uni1k_ident", "uni1k_suffix", "bpe1k", "bpe1k_bf_uds",
          "bpe1k_noesc", "uni32k", "bpe32k", "c5_250k_bf", "bpe1k_llama"]


def fuzz_corpus(n, seed, corpora):
    rng = np.random.default_rng(seed)
    bot = corpora["botchan"][0].tobytes()
    ja = corpora["ja"][0].tobytes().decode("utf-8", errors="ignore")
    out = []
    for i in range(n):
        kind = int(rng.integers(0, 9))
        ln = int(rng.integers(0, 400)) if rng.random() < 0.9 else int(rng.integers(400, 3000))
        if kind == 0:      # byte noise
            s = rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes()
        elif kind == 1:    # mostly ASCII with noise bytes sprinkled in
            b = bytearray(bot[(o := int(rng.integers(0, len(bot) - ln - 1))):o + ln])
            for _ in range(int(rng.integers(0, 4))):
                if b:
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0x80, 0x100))
            s = bytes(b)
        elif kind == 2:    # random code points, all planes (surrogates skipped)
            cps = rng.integers(1, 0x10FFFF, size=ln // 3)
            s = "".join(chr(int(c)) for c in cps if not 0xD800 <= c < 0xE000).encode("utf-8")
        elif kind == 3:    # whitespace storms: ASCII / ideographic / no-break spaces, tabs, U+2581
            al = [" ", "  ", "\t", "　", " ", "▁", "a", "bc", "\n", "\r"]
            s = "".join(al[int(k)] for k in rng.integers(0, len(al), size=ln // 2)).encode("utf-8")
        elif kind == 4:    # Japanese fragment
            o = int(rng.integers(0, max(1, len(ja) - ln - 1)))
            s = ja[o:o + ln // 3].encode("utf-8")
        elif kind == 5:    # compatibility characters the NFKC rules rewrite, in bulk
            al = ["Ａ", "㎒", "ﬁ", "①", "ẛ̣", "Å", "Å", "ｶﾞ", "Ω", "x"]
            s = "".join(al[int(k)] for k in rng.integers(0, len(al), size=ln // 3)).encode("utf-8")
        elif kind == 6:    # one character repeated (long runs of one piece, ties)
            ch = ["a", ".", " ", "猫", "\U0001f600", "\x00"][int(rng.integers(0, 6))]
            s = (ch * (ln // max(1, len(ch.encode())))).encode("utf-8")
        elif kind == 7:    # ASCII words with rule characters between them: replacements that are, begin or end with a
            # space next to real spaces, multi-character replacements, combining marks after ASCII and after kana
            al = ["a", "b", "cd", " ", "　", "\u00a0", "´", "¨", "Ａ", "㍿", "ｶ", "ﾞ", "é", "e\u0301", "か", "\u3099",
                  "①", "猫", "ー", "\u200b", "\U0001f600", "я", "и\u0306"]
            s = "".join(al[int(k)] for k in rng.integers(0, len(al), size=ln // 2)).encode("utf-8")
        else:              # user-defined-symbol lookalikes and reserved names
            al = ["<sep>", "<s>", "</s>", "<unk>", "Botchan", "the end", "...", "<0x41>", " ", "x"]
            s = "".join(al[int(k)] for k in rng.integers(0, len(al), size=ln // 4)).encode("utf-8")
        out.append(s[:8000])
    return synth.pack(out)


@pytest.mark.parametrize("model", MODELS)
def test_fuzz_oracle_vs_reference_and_emulator(model, oracle, corpora):
    blob = fixtures.model_blob(model)
    o = oracle.load(blob)
    text, offs = fuzz_corpus(1500, 99, corpora)
    oids, oio = o.encode_batch(text, offs)
    if refshim.available():
        r = refshim.RefLib().load(blob)
        rids, rio = r.encode_batch(text, offs, threads=8)
        np.testing.assert_array_equal(oio, rio)
        np.testing.assert_array_equal(oids, rids)
        rt, ro = r.decode_batch(rids, rio)
        ot, oo = o.decode_batch(oids, oio)
        np.testing.assert_array_equal(oo, ro)
        np.testing.assert_array_equal(ot, rt)
    from tests import emulib
    e = emulib.EmuLib().load(blob)
    st, so = fixtures.head(text, offs, 250)
    eids, eio = e.encode_batch(st, so, grid=2)
    assert e.status == 0
    np.testing.assert_array_equal(eio, oio[:251])
    np.testing.assert_array_equal(eids, oids[:int(oio[250])])
    et, eo = e.decode_batch(eids, eio, grid=2)
    ot, oo = o.decode_batch(eids, eio)
    np.testing.assert_array_equal(eo, oo)
    np.testing.assert_array_equal(et, ot)


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_fuzz_gpu(model, oracle, corpora):
    from sentencepiece_amd.processor import SentencePieceProcessor
    blob = fixtures.model_blob(model)
    sp = SentencePieceProcessor(model_proto=blob)
    o = oracle.load(blob)
    text, offs = fuzz_corpus(20000, 7, corpora)
    ids, io = sp.EncodePacked(text, offs)
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
    dt, do = sp.DecodePacked(ids, io)
    ot, oo = o.decode_batch(oids, oio)
    np.testing.assert_array_equal(do, oo)
    np.testing.assert_array_equal(dt, ot)
