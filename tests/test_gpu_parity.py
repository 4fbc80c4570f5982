"""Parity tests proper: the HIP path, called through the C ABI, against the
golden ids produced by the compiled reference (tests/golden/) and against the
oracle on the same inputs.  Bit-exact: these are integer ids."""
import numpy as np
import pytest

from tests import fixtures

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def procs():
    from sentencepiece_amd.processor import SentencePieceProcessor
    cache = {}

    def get(model):
        if model not in cache:
            cache[model] = SentencePieceProcessor(model_proto=fixtures.model_blob(model))
        return cache[model]
    return get


def _keys():
    import json
    import os
    with open(os.path.join(fixtures.GOLDEN, "manifest.json")) as f:
        return sorted(k for k in json.load(f) if not k.startswith("_"))


@pytest.mark.parametrize("key", _keys())
def test_golden(key, manifest, golden_arrays, corpora, procs, oracle):
    m = manifest[key]
    sp = procs(m["model"])
    sp.SetEncodeExtraOptions(m["options"])
    text, offs = corpora[m["corpus"]]
    ids, io = sp.EncodePacked(text, offs)
    sp.SetEncodeExtraOptions("")
    cnt = np.diff(io.astype(np.int64))
    gold_cnt = golden_arrays[key + "__cnt"].astype(np.int64)
    bad = np.flatnonzero(cnt != gold_cnt)
    assert bad.size == 0, "sentence %d: %d ids, reference %d" % (bad[0], cnt[bad[0]], gold_cnt[bad[0]])
    if key + "__ids" in golden_arrays:
        np.testing.assert_array_equal(ids, golden_arrays[key + "__ids"])
    assert len(ids) == m["tokens"]
    assert fixtures.sha(ids) == m["sha256"]
    # and the oracle on the same input, element by element
    o = oracle.load(fixtures.model_blob(m["model"]))
    if m["options"]:
        o.set_encode_extra_options(m["options"])
    oids, oio = o.encode_batch(text, offs)
    np.testing.assert_array_equal(io, oio)
    np.testing.assert_array_equal(ids, oids)
