# Model-zoo campaign on CPU (not part of the test suite): fresh models trained with RANDOM trainer / normalizer options
# (the pip sentencepiece wheel is the TRAINER only) -- model type, vocabulary size, byte fallback, dummy prefix, extra
# whitespace kept or removed, whitespace as suffix, whitespace-only pieces, split_by_whitespace / split_digits /
# split_by_unicode_script off, user-defined symbols, the normalization rule, the longest piece -- each loaded into the
# device kernels under the emulator and into the compiled reference (oracle/_ref; the oracle where it is not built), and
# the ids of a mixed corpus compared sentence by sentence: the fuzz corpus of tests/test_fuzz.py (scripts, malformed
# UTF-8, control bytes), word-shaped plain text (scripts/fuzz_plainword.py) and lines of the training text.
# A model is what a user brings: this is the "any .model file" half of the drop-in claim.
# usage: python scripts/fuzz_model_zoo.py SECONDS FIRST_SEED
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from sentencepiece_amd import synth
from tests import fixtures, oraclelib, emulib, wordfuzz
from tests.test_fuzz import fuzz_corpus
import fuzz_plainword

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def training_text(rng, tmp):
    """botchan (English), optionally with the Japanese sample, an indented copy (runs of spaces), digits and symbols."""
    with open(os.path.join(G, "botchan.txt"), "rb") as f:
        lines = f.read().split(b"\n")
    lines = [ln for ln in lines if ln][:int(rng.choice([800, 2000, 4288]))]
    if rng.random() < 0.35:
        with open(os.path.join(G, "ja_sample.txt"), "rb") as f:
            lines += [ln for ln in f.read().split(b"\n") if ln][:300]
    if rng.random() < 0.4:
        lines += [b" " * (2 * (i % 7)) + ln.replace(b" ", b"  " if i % 4 == 0 else b" ") for i, ln in enumerate(lines[:600])]
    if rng.random() < 0.4:
        lines += [b"%d items at $%d.%02d on 20%02d-%02d-%02d #%d" % tuple(int(x) for x in rng.integers(0, 99, size=7)) for _ in range(300)]
    path = os.path.join(tmp, "train.txt")
    with open(path, "wb") as f:
        f.write(b"\n".join(lines) + b"\n")
    return path, lines, len(set(b"\n".join(lines).decode("utf-8", "ignore")))


def random_options(rng):
    o = dict(model_type=str(rng.choice(["unigram", "bpe"])), vocab_size=int(rng.choice([300, 500, 1000, 2000, 4000])),
             normalization_rule_name=str(rng.choice(["nmt_nfkc", "nfkc", "nmt_nfkc_cf", "nfkc_cf", "identity"])),
             character_coverage=float(rng.choice([1.0, 0.9995, 0.98])), hard_vocab_limit=False)
    if rng.random() < 0.4:
        o["byte_fallback"] = True
    if rng.random() < 0.25:
        o["add_dummy_prefix"] = False
    if rng.random() < 0.35:
        o["remove_extra_whitespaces"] = False
    if rng.random() < 0.15:
        o["treat_whitespace_as_suffix"] = True
    if rng.random() < 0.3:
        o["allow_whitespace_only_pieces"] = True
    if rng.random() < 0.15:
        o["split_by_whitespace"] = False
    if rng.random() < 0.3:
        o["split_digits"] = True
    if rng.random() < 0.15:
        o["split_by_unicode_script"] = False
    if rng.random() < 0.15:
        o["split_by_number"] = False
    if rng.random() < 0.3:
        o["max_sentencepiece_length"] = int(rng.choice([4, 8, 24, 40]))
    if rng.random() < 0.25:
        o["user_defined_symbols"] = [str(x) for x in rng.choice(["<sep>", "Botchan", "the end", "...", "▁▁", "http://", "e", "."],
                                                                size=int(rng.integers(1, 4)), replace=False)]
    if rng.random() < 0.1:
        o["control_symbols"] = ["<mask>", "<cls>"]
    if rng.random() < 0.1:
        o["unk_surface"] = "?!"
    return o


def main():
    import sentencepiece as spm
    t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 900)
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    em = emulib.EmuLib()
    try:
        from tests import refshim
        ref = refshim.RefLib() if refshim.available() else None
    except Exception:
        ref = None
    orc = oraclelib.OracleLib()
    corp = fixtures.Corpora()
    bad = n_models = n_sent = n_word = 0
    while time.time() < t_end:
        seed += 1
        rng = np.random.default_rng(seed)
        opts = random_options(rng)
        with tempfile.TemporaryDirectory() as tmp:
            path, lines, n_chars = training_text(rng, tmp)
            # (room for the required characters, the byte pieces and the meta pieces: the trainer refuses less)
            opts["vocab_size"] = max(opts["vocab_size"], n_chars + 300 + (256 if opts.get("byte_fallback") else 0))
            try:
                spm.SentencePieceTrainer.train(input=path, model_prefix=os.path.join(tmp, "m"), num_threads=2, minloglevel=2, **opts)
                with open(os.path.join(tmp, "m.model"), "rb") as f:
                    blob = f.read()
            except Exception as e:
                print("TRAIN", seed, repr(e)[:100], flush=True)      # (an option set the trainer refuses: not ours to judge)
                continue
        try:
            chk = (ref or orc).load(blob)
        except Exception as e:
            print("REFLOAD", seed, opts, repr(e)[:100], flush=True)
            continue
        words = wordfuzz.whole_words(blob) or [b"a", b"the", b"of"]
        t1, o1 = fuzz_corpus(120, seed, corp)
        sents = synth.unpack(t1, o1) + fuzz_plainword.batch(rng, words) + [lines[int(i)] for i in rng.integers(0, len(lines), size=80)]
        text, offs = synth.pack(sents)
        for env, classes in (({}, None), ({}, emulib.SMALL_CLASSES), ({"SPMX_NO_SCAN": "1", "SPMX_NO_IDS16": "1"}, None)):
            try:
                h = em.load(blob, cus=2, classes=classes, env=env)
                ids, io = h.encode_batch(text, offs)
                oi, oo = chk.encode_batch(text, offs, threads=2) if ref is not None else chk.encode_batch(text, offs)
                k = wordfuzz.first_difference(np.asarray(ids), np.asarray(io), np.asarray(oi), np.asarray(oo))
                if h.status or k >= 0:
                    bad += 1
                    print("MISMATCH seed", seed, opts, env, "small" if classes else "default", "sentence", k,
                          repr(sents[k][:80]) if k >= 0 else "", "status", h.status, flush=True)
                n_word += sum(c["sentences"] for c in h.sp.LastProfile()["classes"] if c["kernel"].startswith("EncodeWord"))
                n_sent += len(sents)
            except Exception as e:
                bad += 1
                print("EXC seed", seed, opts, env, repr(e)[:200], flush=True)
        # extra options (bos / eos / reverse) and a restricted vocabulary (SetVocabulary: pieces outside it are resegmented or
        # become unused, src/sentencepiece_processor.cc:339-372), the reference's own switches on both sides
        try:
            h = em.load(blob, cus=2)
            from sentencepiece import sentencepiece_model_pb2 as pb
            mp = pb.ModelProto()
            mp.ParseFromString(blob)
            normal = [p.piece for p in mp.pieces if p.type == 1]
            keep = [normal[int(i)] for i in rng.choice(len(normal), size=max(1, len(normal) // int(rng.choice([2, 3, 10]))), replace=False)]
            for step in ("opts", "vocab", "reset"):
                if step == "opts":
                    x = str(rng.choice(["bos", "eos", "bos:eos", "reverse", "reverse:bos:eos"]))
                    h.set_encode_extra_options(x); chk.set_encode_extra_options(x)
                elif step == "vocab":
                    h.set_vocabulary(keep); chk.set_vocabulary(keep)
                else:
                    h.reset_vocabulary(); chk.reset_vocabulary()
                ids, io = h.encode_batch(text, offs)
                oi2, oo2 = chk.encode_batch(text, offs, threads=2) if ref is not None else chk.encode_batch(text, offs)
                k = wordfuzz.first_difference(np.asarray(ids), np.asarray(io), np.asarray(oi2), np.asarray(oo2))
                if h.status or k >= 0:
                    bad += 1
                    print("MISMATCH(%s) seed" % step, seed, opts, "sentence", k, repr(sents[k][:80]) if k >= 0 else "", "status", h.status, flush=True)
            chk.set_encode_extra_options("")
        except Exception as e:
            bad += 1
            print("EXC(options) seed", seed, opts, repr(e)[:200], flush=True)
        # the other entry points of the path's neighbourhood against the ORACLE (which is itself compared with the compiled
        # reference on the same input first: a model the restatement gets wrong must show up as that, not as a kernel bug)
        try:
            o = orc.load(blob)
            xi, xo = o.encode_batch(text, offs)
            if ref is not None and wordfuzz.first_difference(np.asarray(xi), np.asarray(xo), np.asarray(oi), np.asarray(oo)) >= 0:
                bad += 1
                print("ORACLE != REFERENCE seed", seed, opts, flush=True)
            h = em.load(blob, cus=2)
            short = [s for s in sents if len(s) <= 4000]
            t2, o2 = synth.pack(short)
            got, want = h.encode_spans(t2, o2), o.encode_spans(t2, o2)
            if h.status or not all(np.array_equal(np.asarray(a).astype(np.int64), np.asarray(b).astype(np.int64)) for a, b in zip(got, want)):
                bad += 1
                print("SPANS MISMATCH seed", seed, opts, flush=True)
            a, b = h.normalize_batch(t2, o2), o.normalize_batch(t2, o2)
            if not all(np.array_equal(x, y) for x, y in zip(a, b)):
                bad += 1
                print("NORMALIZE MISMATCH seed", seed, opts, flush=True)
            di, do = o.encode_batch(t2, o2)
            dt, dd = h.decode_batch(di, do)
            et, ed = o.decode_batch(di, do)
            if not (np.array_equal(dt, et) and np.array_equal(dd, ed)):
                bad += 1
                print("DECODE MISMATCH seed", seed, opts, flush=True)
            # NBestEncode of short sentences (unigram models): ids and scores against the compiled reference's own
            if opts["model_type"] == "unigram" and ref is not None:
                from tests.test_nbest import nbest as nb_call
                few = [s for s in sents if 0 < len(s) <= 60][:24]
                if few:
                    t3, o3 = synth.pack(few)
                    kk = int(rng.choice([1, 2, 5, 9]))
                    res = h.nbest(t3, o3, kk)
                    for sent, r in zip(few, res):
                        n, want, sc = nb_call(chk.lib.spmref_nbest_encode, chk.h, sent, kk)
                        if n < 0:
                            continue
                        if [x[0] for x in r] != want or not np.array_equal(np.array([x[1] for x in r], dtype=np.float32), sc):
                            bad += 1
                            print("NBEST MISMATCH seed", seed, opts, kk, repr(sent[:60]), flush=True)
                            break
        except Exception as e:
            bad += 1
            print("EXC(other entry points) seed", seed, opts, repr(e)[:200], flush=True)
        n_models += 1
        if n_models % 10 == 0:
            print("seed", seed, "models", n_models, "bad", bad, "sentence encodings", n_sent, "word-form share %.3f" % (n_word / max(1, n_sent)), flush=True)
    print("DONE models", n_models, "bad", bad, "sentence encodings", n_sent, "through the word form", n_word)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
