"""Shared test inputs: the deterministic corpora the golden ids were made on
(scripts/make_fixtures.py builds exactly these) and model blobs."""
import os

import numpy as np

from scripts import make_fixtures as mf

GOLDEN = mf.G


def model_blob(name):
    """Serialized ModelProto of a fixture model.  The 250k-piece C5 models (BASELINE.json configs[4]) are
    synthesized deterministically (sentencepiece_amd/synth.py c5_model) and cached under the temp dir
    instead of being stored in the repository (4.7 MB each)."""
    if name in ("c5_250k", "c5_250k_bf"):
        import tempfile
        from sentencepiece_amd import synth
        path = os.path.join(tempfile.gettempdir(), "spmx_%s_v1.model" % name)
        if not os.path.exists(path):
            blob = synth.c5_model(model_blob("uni32k"), byte_fallback=name.endswith("_bf"))
            with open(path + ".tmp%d" % os.getpid(), "wb") as f:
                f.write(blob)
            os.replace(path + ".tmp%d" % os.getpid(), path)
        with open(path, "rb") as f:
            return f.read()
    with open(os.path.join(GOLDEN, name + ".model"), "rb") as f:
        return f.read()


class Corpora:
    """Lazy name -> (text uint8, offsets uint64)."""

    def __init__(self):
        self._c = {}

    def __getitem__(self, name):
        if name not in self._c:
            if not self._c.get("_all"):
                self._c.update(mf.corpora())
                self._c["_all"] = True
        return self._c[name]


def sha(ids):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(ids).astype("<i4").tobytes()).hexdigest()


def head(text, offs, n):
    """First n sentences of a packed buffer."""
    n = min(n, len(offs) - 1)
    return text[:int(offs[n])], offs[:n + 1]
