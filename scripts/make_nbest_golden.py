"""Golden vectors for the piece / proto forms of NBestEncode (tests/golden/nbest_protos.json): the serialized
NBestSentencePieceText the REFERENCE produces (the `sentencepiece` Python module of this image, v0.2.2 -- the same
sentencepiece_processor.cc:653-676 path as /root/reference) for a few sentences per fixture model, no extra options.  Run in the build container; the GPU box only reads the JSON."""
import json
import os
import sys

import sentencepiece as spm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")

SENTENCES = [
    "hello world", "", " ", "  leading and trailing  ", "Hello, World! 123", "a", "the quick brown fox jumps over the lazy dog",
    "ＡＢＣ　ｄｅｆ", "猫が好き", "naïve café", "x y", "\U0001f600\U0001f600 ok \U0001f600", "日本語のテキスト, mixed with English.",
    "<s> </s> <unk>", "éé", "㍿", "tab\there", "▁already▁escaped",
]
# (the module does not expose SetEncodeExtraOptions: the extra options are covered by tests/test_lattice_pieces.py against
# the spans form of the plain encoder, which has reference-made golden vectors of its own)
CASES = [("test_model", ""), ("uni1k_bf", ""), ("uni1k_uds", ""), ("test_ja_model", ""), ("uni1k_ident", ""),
         ("uni1k_suffix", ""), ("uni32k", "")]


def main():
    out = {"_what": "serialized NBestSentencePieceText per (model, extra options, nbest_size, sentence), hex; made by scripts/make_nbest_golden.py "
                    "with sentencepiece " + spm.__version__, "sentences": SENTENCES, "cases": []}
    for model, opts in CASES:
        sp = spm.SentencePieceProcessor(model_file=os.path.join(G, model + ".model"))
        for nbest in (1, 2, 5):
            blobs = [sp.NBestEncodeAsSerializedProto(s, nbest).hex() for s in SENTENCES]
            out["cases"].append({"model": model, "options": opts, "nbest": nbest, "protos": blobs})
    with open(os.path.join(G, "nbest_protos.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    sys.exit(main())
