"""Campaign for the lane normalizers (kernels_normlane.h) under the emulator against the oracle: per seed
2000 fuzz sentences (tests/test_fuzz.py kinds, incl. rule characters between ASCII words) and 600 mixed-script ones, nine
models, three normalizer settings (character-stepping form forced with / without the word kernels, automatic), two class
tables.  usage: python scripts/fuzz_normalizer_forms.py SEED [SEED ...]   (round 3: seeds 1-3 and 11-30, 0 differences)"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import fixtures, emulib, oraclelib, test_fuzz
from sentencepiece_amd import synth
C=fixtures.Corpora()
E=emulib.EmuLib(); O=oraclelib.OracleLib()
seeds=[int(x) for x in sys.argv[1:]] or [1]
for seed in seeds:
    text,offs=test_fuzz.fuzz_corpus(2000, seed, C)
    t2,o2=synth.mixed_corpus(600, seed=seed, hi=2500)
    for model in ("test_model","uni1k_bf","c5_250k","bpe1k_noesc","uni1k_ident","bpe32k","test_ja_model","uni32k","bpe1k_llama"):
        blob=fixtures.model_blob(model)
        o=O.load(blob)
        for (t,of) in ((text,offs),(t2,o2)):
            rid,rio=o.encode_batch(t,of)
            for env in ({"SPMX_CHAR_NORM_ALWAYS":"1","SPMX_NO_WORD_KERNEL":"1"},{},{"SPMX_CHAR_NORM_ALWAYS":"1"}):
                for classes in ("", emulib.SMALL_CLASSES):
                    h=E.load(blob, classes=classes, env=dict(env))
                    ids,io=h.encode_batch(t,of)
                    ok=np.array_equal(ids,rid) and np.array_equal(io,rio)
                    if not ok:
                        print("MISMATCH", seed, model, env, bool(classes)); 
                        iol=io.astype(np.int64); riol=rio.astype(np.int64)
                        for i in range(len(of)-1):
                            if not np.array_equal(ids[iol[i]:iol[i+1]], rid[riol[i]:riol[i+1]]):
                                print(i, bytes(t[int(of[i]):int(of[i+1])])[:200]); break
                        sys.exit(1)
        print(seed, model, "ok"); sys.stdout.flush()
