#!/usr/bin/env python3
"""Trains tests/golden/uni32k_w16.model: the C2 recipe (32k unigram, nmt_nfkc, the synthetic ASCII generator) on a word
list whose words have up to 16 letters (mean 5.5), as natural text with the trainer's default
max_sentencepiece_length = 16 gives: the longest piece then has 17 bytes, so the engine's score ring has 18 entries and
the generic streaming kernel runs instead of the 16-entry specialization.  bench.py --model uni32k_w16 reports that
rate next to the headline's.  Uses the pip wheel's trainer (training is not on the device path)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sentencepiece_amd import synth  # noqa: E402
from scripts.make_fixtures import train  # noqa: E402


def main():
    words = synth.WordList(max_word_len=16, mean_word_len=5.5)
    text, offs = synth.ascii_corpus(300_000, seed=777, words=words)
    with tempfile.NamedTemporaryFile("wb", suffix=".txt", delete=False) as f:
        for s in synth.unpack(text, offs):
            if b"\n" in s or b"\r" in s:
                continue
            f.write(s + b"\n")
        sample = f.name
    train("uni32k_w16", sample, model_type="unigram", vocab_size=32000, normalization_rule_name="nmt_nfkc",
          input_sentence_size=300000, shuffle_input_sentence=False, hard_vocab_limit=False,
          train_extremely_large_corpus=False, max_sentence_length=8192)
    os.remove(sample)


if __name__ == "__main__":
    main()
