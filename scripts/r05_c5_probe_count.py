# CPU analysis for DESIGN section 6 item 3 (round 4): dependent trie probes per character of the C5 workload under three first-step
# tables (first byte in LDS = today; first character; first two characters).  usage: python scripts/r05_c5_probe_count.py
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, collections
from sentencepiece_amd import synth
from tests import fixtures
from sentencepiece import sentencepiece_model_pb2 as pb
import sentencepiece as spm
blob = fixtures.model_blob("c5_250k")
m = pb.ModelProto(); m.ParseFromString(blob)
sp = spm.SentencePieceProcessor(model_proto=blob)
# byte trie as nested dict
root = {}
for p in m.pieces:
    if p.type not in (1, 4, 6): continue
    node = root
    for b in p.piece.encode():
        node = node.setdefault(b, {})
    node[-1] = True
text, offs = synth.mixed_corpus(3000, seed=20250228)
tb = text.tobytes()
tot_bytes = tot_chars = 0
p_byte = p_char = p_char2 = 0      # HBM probes: first byte from LDS (today) / first character from a table / first two characters
steps_hist = collections.Counter()
for i in range(len(offs) - 1):
    s = tb[int(offs[i]):int(offs[i+1])]
    try:
        norm = sp.normalize(s.decode("utf-8", "ignore")).encode()
    except Exception:
        continue
    n = len(norm); tot_bytes += n
    pos = 0
    while pos < n:
        c = norm[pos]
        mb = 1 if c < 0x80 else (2 if c < 0xE0 else (3 if c < 0xF0 else 4))
        mb = min(mb, n - pos)
        tot_chars += 1
        node = root; d = 0
        while pos + d < n and norm[pos + d] in node:
            node = node[norm[pos + d]]; d += 1
        # d = matched bytes; probes issued: one per matched byte + one failing probe (the child bitmap prunes most failing ones: count both)
        steps_hist[d] += 1
        p_byte += max(0, d - 1)            # byte 0 from the root table in LDS; matched bytes 1 .. d-1 are HBM probes (failing probe pruned by the bitmap)
        p_char += max(0, d - mb)           # bytes of the first character from a code-point table
        # second character
        mb2 = 0
        if pos + mb < n:
            c2 = norm[pos + mb]; mb2 = 1 if c2 < 0x80 else (2 if c2 < 0xE0 else (3 if c2 < 0xF0 else 4))
        p_char2 += max(0, d - mb - mb2)
        pos += mb
print("normalized bytes", tot_bytes, "characters", tot_chars, "bytes/char %.2f" % (tot_bytes / tot_chars))
print("HBM probes per character: first BYTE in LDS (today) %.2f; first CHARACTER from a table %.2f; first two characters %.2f" % (p_byte / tot_chars, p_char / tot_chars, p_char2 / tot_chars))
print("matched depth histogram (bytes):", sorted(steps_hist.items())[:16])
