"""include/spmx_processor.h (the C++ SentencePieceProcessor-shaped facade over
the C ABI), compiled with g++ and linked against libspmx.so."""
import os
import subprocess

import numpy as np
import pytest

from tests import fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "facade_test")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "facade_test.cc")
    lib = os.path.join(ROOT, "sentencepiece_amd")
    if (not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(src)
            or os.path.getmtime(BIN) < os.path.getmtime(os.path.join(ROOT, "include", "spmx_processor.h"))):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", BIN, src, "-L" + lib, "-lspmx",
                               "-Wl,-rpath," + lib])
    return BIN


def test_facade_builds_and_reports_unavailable():
    import torch
    b = _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([b, os.path.join(fixtures.GOLDEN, "test_model.model"), "--expect-unavailable"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "Unavailable" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("model,opts,key", [("test_model", "", "test_model__botchan"),
                                             ("test_model", "bos:eos", "test_model__botchan__bos-eos"),
                                             ("bpe1k", "", "bpe1k__botchan")])
def test_facade_matches_golden(model, opts, key, golden_arrays, tmp_path):
    b = _build()
    args = [b, os.path.join(fixtures.GOLDEN, model + ".model"), os.path.join(fixtures.GOLDEN, "botchan.txt")]
    if opts:
        args.append(opts)
    out = subprocess.run(args, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ids = np.array([int(x) for line in out.stdout.split("\n") for x in line.split()], dtype=np.int32)
    np.testing.assert_array_equal(ids, golden_arrays[key + "__ids"])
