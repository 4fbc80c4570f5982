#!/usr/bin/env python3
"""Generates tests/golden/: model fixtures, text fixtures and golden token ids.

Runs only in the build container (needs /root/reference for the two bundled
models + corpora, the pip `sentencepiece` wheel for *training* new models, and
oracle/_ref/libspm_ref.so -- the compiled reference -- for the golden ids).
The wheel is never used as a parity oracle (SURVEY.md finding 4): every golden
id below comes from the reference compiled from /root/reference.

    python scripts/make_fixtures.py            # everything
    python scripts/make_fixtures.py --only ids # just re-generate golden ids
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sentencepiece_amd import synth  # noqa: E402
from tests import refshim  # noqa: E402

REF = "/root/reference"
G = os.path.join(ROOT, "tests", "golden")


def train(name, corpus_path, **kw):
    import sentencepiece as spm
    out = os.path.join(G, name)
    args = dict(input=corpus_path, model_prefix=out, num_threads=8,
                minloglevel=2)
    args.update(kw)
    spm.SentencePieceTrainer.train(**args)
    os.remove(out + ".vocab")
    print("trained", name, os.path.getsize(out + ".model"))


def make_models():
    os.makedirs(G, exist_ok=True)
    shutil.copy(f"{REF}/python/test/test_model.model", f"{G}/test_model.model")
    shutil.copy(f"{REF}/python/test/test_ja_model.model", f"{G}/test_ja_model.model")
    shutil.copy(f"{REF}/data/botchan.txt", f"{G}/botchan.txt")
    with open(f"{REF}/data/wagahaiwa_nekodearu.txt", "rb") as f:
        lines = f.read().split(b"\n")[:700]
    # paragraphs of up to 17 KB: cut them at character boundaries so that every
    # fixture line fits the device path's largest common length class (<= 4096 B)
    cut = []
    for ln in lines:
        s = ln.decode("utf-8")
        while len(s.encode("utf-8")) > 3900:
            k = 1300
            cut.append(s[:k].encode("utf-8"))
            s = s[k:]
        cut.append(s.encode("utf-8"))
    with open(f"{G}/ja_sample.txt", "wb") as f:
        f.write(b"\n".join(cut) + b"\n")
    bot = f"{G}/botchan.txt"
    train("uni1k", bot, vocab_size=1000, model_type="unigram")
    train("bpe1k", bot, vocab_size=1000, model_type="bpe")
    train("uni1k_bf", bot, vocab_size=1000, model_type="unigram", byte_fallback=True,
          character_coverage=0.98)
    train("bpe1k_bf_uds", bot, vocab_size=1000, model_type="bpe", byte_fallback=True,
          character_coverage=0.98,
          user_defined_symbols=["<sep>", "Botchan", "the end", "▁▁"])
    train("uni1k_uds", bot, vocab_size=1000, model_type="unigram",
          user_defined_symbols=["<sep>", "Botchan", "the end", "..."])
    train("uni1k_ident", bot, vocab_size=1000, model_type="unigram",
          normalization_rule_name="identity", add_dummy_prefix=False,
          remove_extra_whitespaces=False)
    train("uni1k_suffix", bot, vocab_size=1000, model_type="unigram",
          normalization_rule_name="nfkc_cf", treat_whitespace_as_suffix=True)
    train("bpe1k_noesc", bot, vocab_size=1000, model_type="bpe",
          normalization_rule_name="nmt_nfkc_cf", split_by_whitespace=False)
    # Llama-style BPE: identity normalizer, no whitespace removal, byte fallback, whitespace-only pieces (runs of
    # U+2581), digits split -- trained on botchan plus an indented copy of it so that space runs are frequent
    with open(bot, "rb") as f:
        lines = f.read().split(b"\n")
    with tempfile.NamedTemporaryFile("wb", suffix=".txt", delete=False) as f:
        for i, ln in enumerate(lines):
            f.write(ln + b"\n")
            f.write(b" " * (2 * (i % 9)) + ln.replace(b" ", b"  " if i % 5 == 0 else b" ") + b"\n")
        llama_in = f.name
    train("bpe1k_llama", llama_in, vocab_size=1000, model_type="bpe", byte_fallback=True, character_coverage=0.98,
          normalization_rule_name="identity", remove_extra_whitespaces=False, allow_whitespace_only_pieces=True,
          split_digits=True, add_dummy_prefix=True)
    os.remove(llama_in)
    # configs 2 / 3: 32k unigram + BPE on a sample of the synthetic generator
    text, offs = synth.ascii_corpus(300_000, seed=777)
    with tempfile.NamedTemporaryFile("wb", suffix=".txt", delete=False) as f:
        for s in synth.unpack(text, offs):
            if b"\n" in s or b"\r" in s:
                continue
            f.write(s + b"\n")
        sample = f.name
    common = dict(vocab_size=32000, normalization_rule_name="nmt_nfkc",
                  input_sentence_size=300000, shuffle_input_sentence=False,
                  hard_vocab_limit=False, train_extremely_large_corpus=False,
                  max_sentence_length=8192)
    train("uni32k", sample, model_type="unigram", **common)
    train("bpe32k", sample, model_type="bpe", **common)
    os.remove(sample)


def corpora():
    """name -> (text, offs).  Deterministic; regenerated (not stored) by tests."""
    out = {}
    with open(f"{G}/botchan.txt", "rb") as f:
        # spm_encode semantics: getline strips '\n' only ('\r' is kept).
        lines = f.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    out["botchan"] = synth.pack(lines)
    with open(f"{G}/ja_sample.txt", "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    out["ja"] = synth.pack(lines)
    out["synth20k"] = synth.ascii_corpus(20_000, seed=20250227, workers=1)
    out["mixed2k"] = synth.mixed_corpus(2_000, seed=20250228)
    out["edge"] = synth.pack(edge_sentences())
    return out


def edge_sentences():
    """Hand-written edge cases mirroring the reference's unit tests
    (normalizer_test.cc:37-147, :266-275; bpe_model_test.cc:183-186;
    sentencepiece_processor_test.cc:186-303)."""
    return [
        b"", b" ", b"   ", b"\t", b"a", b" a", b"a ", b"  a  b  ", b"a\tb", b"\r", b"a\r\n",
        "　　".encode(), "　a　".encode(), "▁".encode(), "▁▁a▁▁".encode(),
        "a▁".encode(), " ▁ ".encode(), "ＡＢＣ".encode(), "㍿".encode(), "①②".encode(),
        "ｶﾞｷﾞ".encode(), "ﷺ".encode(), "ǆ".encode(), "Å".encode(), "éé".encode(),
        "­".encode(), "a­b".encode(), "​".encode(), " ​ ".encode(),
        b"\x80", b"\xff\xfe", b"\xe2\x96", b"\xe2", b"ab\xc0\xafcd", b"\xed\xa0\x80", b"\xf4\x90\x80\x80",
        b"\xef\xbf\xbd", b"a\xef\xbf\xbdb", b"\xf0\x9f\x98\x80", b"\xc2", b"\xc2\xa0", b"x\x00y", b"\x00",
        b"......", b". .....", b"I saw a girl with a telescope.", b"<sep>", b"a<sep>b <sep> c", b"Botchan",
        b"the end", b"thethe end the  end", b"...", b"....", b"Hello World!!", b"hello  world",
        " hello  world ".encode(), "吾輩は猫である。名前はまだ無い。".encode(),
        "これはテストです ABC 123".encode(), "𠮷野家".encode(), "💩💩💩".encode(),
        b"a" * 300, b"ab " * 200, ("猫" * 150).encode(), b" " * 100 + b"x" + b" " * 100,
        b"0123456789" * 40, ("ＡＢ " * 120).encode(),
    ]


def decode_fuzz_ids(vocab_size, seed=20250301):
    """Seeded random CSR ids for the Decode parity tests: 600 sentences of 0..150 ids, two thirds of them drawn
    from the first 300 ids (where the control, unknown and byte pieces live)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 150, size=600)
    total = int(lens.sum())
    low = rng.integers(0, min(vocab_size, 300), size=total)
    high = rng.integers(0, vocab_size, size=total)
    ids = np.where(rng.random(total) < 0.67, low, high).astype(np.int32)
    io = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    return ids, io


MODELS = ["test_model", "test_ja_model", "uni1k", "bpe1k", "uni1k_bf", "bpe1k_bf_uds", "uni1k_uds",
          "uni1k_ident", "uni1k_suffix", "bpe1k_noesc", "uni32k", "bpe32k", "bpe1k_llama"]
PAIRS = [(m, c) for m in MODELS for c in ("botchan", "edge", "mixed2k")] + \
        [("test_ja_model", "ja"), ("uni1k_bf", "ja"), ("bpe1k_bf_uds", "ja"),
         ("uni32k", "synth20k"), ("bpe32k", "synth20k"), ("test_model", "synth20k"),
         # BASELINE.json configs[4]: 250k-piece unigram (synthesized, see synth.c5_model) on mixed-script text
         ("c5_250k", "mixed2k"), ("c5_250k_bf", "mixed2k"), ("c5_250k", "edge"), ("c5_250k_bf", "edge"),
         ("c5_250k", "ja")]
EXTRA = [("test_model", "botchan", "bos:eos"), ("test_model", "botchan", "reverse:bos"),
         ("bpe1k", "edge", "eos:reverse:bos")]


def make_ids():
    ref = refshim.RefLib()
    cs = corpora()
    manifest = {}
    arrays = {}
    for m, c, *opt in [p + ("",) if len(p) == 2 else p for p in PAIRS + EXTRA]:
        opt = opt[0] if opt else ""
        from tests import fixtures
        blob = fixtures.model_blob(m)
        h = ref.load(blob)
        if opt:
            h.set_encode_extra_options(opt)
        text, offs = cs[c]
        ids, id_offs = h.encode_batch(text, offs, threads=8)
        key = f"{m}__{c}" + (f"__{opt.replace(':', '-')}" if opt else "")
        # full ids only for the small corpora; per-sentence counts + a digest
        # of the id stream for everything (keeps tests/golden small).
        if c in ("edge", "botchan", "ja") and len(ids) <= 130_000:
            arrays[key + "__ids"] = ids.astype(np.int32)
        arrays[key + "__cnt"] = np.diff(id_offs.astype(np.int64)).astype(np.uint16)
        manifest[key] = dict(model=m, corpus=c, options=opt, n=int(len(offs) - 1), tokens=int(len(ids)),
                             sha256=hashlib.sha256(ids.astype("<i4").tobytes()).hexdigest())
        # Decode(ids) of the same ids by the compiled reference: bytes + digest of (text, offsets)
        dtext, doffs = h.decode_batch(ids, id_offs)
        manifest[key]["decode_bytes"] = int(len(dtext))
        manifest[key]["decode_sha256"] = hashlib.sha256(dtext.tobytes() + doffs.astype("<u8").tobytes()).hexdigest()
        print(key, manifest[key]["tokens"])
    # Decode of seeded random id sequences (byte pieces in any order, control / unknown pieces, empty sentences)
    fuzz = {}
    for m in sorted(set(p[0] for p in PAIRS)):
        from tests import fixtures
        h = ref.load(fixtures.model_blob(m))
        ids, io = decode_fuzz_ids(h.lib.spmref_piece_size(h.h))
        dtext, doffs = h.decode_batch(ids, io)
        fuzz[m] = dict(bytes=int(len(dtext)),
                       sha256=hashlib.sha256(dtext.tobytes() + doffs.astype("<u8").tobytes()).hexdigest())
    manifest["_decode_fuzz"] = fuzz
    np.savez_compressed(f"{G}/golden_ids.npz", **arrays)
    with open(f"{G}/manifest.json", "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


def spans_digest(b, e):
    return hashlib.sha256(np.asarray(b).astype("<u4").tobytes() + np.asarray(e).astype("<u4").tobytes()).hexdigest()


def make_spans():
    """Adds to every pair of the manifest the digest of pieces(i).begin / .end that the compiled reference's
    Encode(input, SentencePieceText*) gives (the ids of that call are checked against the stored digest)."""
    from tests import fixtures
    ref = refshim.RefLib()
    cs = corpora()
    with open(f"{G}/manifest.json") as f:
        manifest = json.load(f)
    for key, m in sorted(manifest.items()):
        if key.startswith("_"):
            continue
        h = ref.load(fixtures.model_blob(m["model"]))
        if m["options"]:
            h.set_encode_extra_options(m["options"])
        text, offs = cs[m["corpus"]]
        ids, b, e, io = h.encode_spans(text, offs)
        assert hashlib.sha256(ids.astype("<i4").tobytes()).hexdigest() == m["sha256"], key
        m["spans_sha256"] = spans_digest(b, e)
        print(key, m["spans_sha256"][:12])
    with open(f"{G}/manifest.json", "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["models", "ids", "spans"], default=None)
    a = ap.parse_args()
    if a.only in (None, "models"):
        make_models()
    if a.only in (None, "ids"):
        make_ids()
    if a.only in (None, "spans"):
        make_spans()
