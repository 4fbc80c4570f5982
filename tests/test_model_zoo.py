"""A slice of the model-zoo campaign (scripts/fuzz_model_zoo.py) as tests: models TRAINED ON THE SPOT with random trainer /
normalizer options (unigram / BPE, byte fallback, no dummy prefix, extra whitespace kept, whitespace as suffix,
whitespace-only pieces, split switches off, user-defined and control symbols, five normalization rules, longest piece
4 - 40), each loaded into the product and into the compiled reference (oracle/_ref; the oracle where it is not built):
encode (ids of every sentence), the spans form and Decode compared.  "Any .model file a user brings" is a claim about
HARDWARE, so the large leg is -m gpu (>= 50 models through the HIP path and the C ABI); a small emulated leg keeps the
harness itself honest on CPU.  The pip sentencepiece wheel is the TRAINER only (SURVEY finding 4)."""
import os
import sys
import tempfile

import numpy as np
import pytest

from sentencepiece_amd import synth
from tests import fixtures, wordfuzz
from tests.test_fuzz import fuzz_corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _train(seed):
    """-> (model blob, options, training lines) or None when the trainer refuses the option set."""
    import sentencepiece as spm
    import fuzz_model_zoo as zoo
    rng = np.random.default_rng(seed)
    opts = zoo.random_options(rng)
    opts["vocab_size"] = min(opts["vocab_size"], 1000)
    with tempfile.TemporaryDirectory() as tmp:
        path, lines, n_chars = zoo.training_text(rng, tmp)
        opts["vocab_size"] = max(opts["vocab_size"], n_chars + 300 + (256 if opts.get("byte_fallback") else 0))
        try:
            spm.SentencePieceTrainer.train(input=path, model_prefix=os.path.join(tmp, "m"), num_threads=2, minloglevel=2, **opts)
            with open(os.path.join(tmp, "m.model"), "rb") as f:
                return f.read(), opts, lines, rng
        except Exception:
            return None


def _check_model(encode, spans, decode, blob, opts, lines, rng, seed, corpora, oracle):
    """encode / spans / decode: the product's packed entry points.  Returns the number of sentences compared."""
    import fuzz_plainword
    from tests import refshim
    ref = refshim.RefLib().load(blob) if refshim.available() else None
    o = oracle.load(blob)
    words = wordfuzz.whole_words(blob) or [b"a", b"the", b"of"]
    t1, o1 = fuzz_corpus(60, seed, corpora)
    sents = synth.unpack(t1, o1) + fuzz_plainword.batch(rng, words)[:150] + [lines[int(i)] for i in rng.integers(0, len(lines), size=60)]
    text, offs = synth.pack(sents)
    ids, io = encode(text, offs)
    wi, wo = ref.encode_batch(text, offs, threads=2) if ref is not None else o.encode_batch(text, offs)
    k = wordfuzz.first_difference(np.asarray(ids), np.asarray(io), np.asarray(wi), np.asarray(wo))
    assert k < 0, ("encode", seed, opts, k, sents[k][:80])
    if ref is not None:           # (the oracle pinned to the reference on this very model before it checks the other forms)
        xi, xo = o.encode_batch(text, offs)
        assert wordfuzz.first_difference(np.asarray(xi), np.asarray(xo), np.asarray(wi), np.asarray(wo)) < 0, ("oracle != reference", seed, opts)
    short = [s for s in sents if len(s) <= 4000]
    t2, o2 = synth.pack(short)
    got, want = spans(t2, o2), o.encode_spans(t2, o2)
    assert all(np.array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64)) for a, b in zip(got, want)), ("spans", seed, opts)
    di, do = o.encode_batch(t2, o2)
    dt, dd = decode(di, do)
    et, ed = (ref.decode_batch(di, do) if ref is not None else o.decode_batch(di, do))
    assert np.array_equal(np.asarray(dd).astype(np.int64), np.asarray(ed).astype(np.int64)) and np.array_equal(dt, et), ("decode", seed, opts)
    return len(sents)


def test_emu_model_zoo_slice(oracle, corpora):
    from tests import emulib
    em = emulib.EmuLib()
    done = 0
    for seed in range(91000, 91012):
        m = _train(seed)
        if m is None:
            continue
        blob, opts, lines, rng = m
        h = em.load(blob, cus=2, classes=None)
        _check_model(h.encode_batch, h.encode_spans, h.decode_batch, blob, opts, lines, rng, seed, corpora, oracle)
        done += 1
        if done >= 4:
            break
    assert done >= 3


@pytest.mark.gpu
def test_gpu_model_zoo_fifty_models(oracle, corpora):
    from sentencepiece_amd.processor import SentencePieceProcessor
    done = n_sent = 0
    seed = 92000
    while done < 50 and seed < 92200:
        seed += 1
        m = _train(seed)
        if m is None:
            continue
        blob, opts, lines, rng = m
        sp = SentencePieceProcessor(model_proto=blob)
        n_sent += _check_model(sp.EncodePacked, sp.EncodeSpansPacked, sp.DecodePacked, blob, opts, lines, rng, seed, corpora, oracle)
        done += 1
        del sp
    assert done >= 50, done
    print("model zoo on the GPU: %d models, %d sentence encodings compared" % (done, n_sent))
