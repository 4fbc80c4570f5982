"""The device corpus packer (sentencepiece_amd/csrc/kernels_split.h) against std::getline semantics, the loop of
the reference's spm_encode (src/spm_encode_main.cc:159-165).  CPU: the device body under the wavefront emulator;
GPU: through the C ABI."""
import os
import random

import numpy as np
import pytest

from tests.emulib import EmuLib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def getline_split(data):
    """What a std::getline loop yields: '\\n' ends a line, a trailing line without it counts, "a\\n" is one line."""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    return lines


def packed(lines):
    offs = np.zeros(len(lines) + 1, dtype=np.uint64)
    if lines:
        offs[1:] = np.cumsum([len(x) for x in lines])
    return b"".join(lines), offs


def cases():
    rng = random.Random(5)
    out = [b"", b"\n", b"a", b"a\n", b"a\nb", b"\n\n\n", b"\na\n\nb\n", b"x" * 15 + b"\n", b"x" * 16 + b"\n", b"x" * 17,
           b"a\r\nb\r\n", b"\n" * 1024, b"\n" * 1025 + b"z", b"y" * 1023 + b"\n", b"y" * 1024 + b"\n" + b"y" * 1024,
           b"q" * 16383 + b"\n", b"q" * 16384 + b"\n", b"q" * 16385, ("▁あ\n" * 700).encode()]
    for n in (1000, 5000, 16384 * 2 + 5, 70001):
        for p_nl in (0.01, 0.1, 0.6):
            out.append(bytes(0x0A if rng.random() < p_nl else rng.choice(b"ab \r\xe3\x81\x82") for _ in range(n)))
    return out


@pytest.fixture(scope="module")
def emu():
    return EmuLib()


@pytest.mark.parametrize("grid", [1, 3])
def test_split_emulated(emu, grid):
    for data in cases():
        text, offs = emu.split_lines(data, grid=grid)
        want_text, want_offs = packed(getline_split(data))
        assert text == want_text, (len(data), data[:40])
        assert np.array_equal(offs, want_offs), (len(data), data[:40])


def test_split_emulated_botchan(emu):
    path = os.path.join(ROOT, "tests", "golden", "botchan.txt")
    if not os.path.exists(path):
        pytest.skip("no corpus fixture")
    data = open(path, "rb").read()
    text, offs = emu.split_lines(data, grid=5)
    want_text, want_offs = packed(getline_split(data))
    assert text == want_text and np.array_equal(offs, want_offs)


@pytest.mark.gpu
def test_split_gpu():
    import torch
    from tests import fixtures
    from sentencepiece_amd.processor import SentencePieceProcessor
    sp = SentencePieceProcessor(model_proto=fixtures.model_blob("uni1k"), device=0)
    for data in cases():
        d_file = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda:0")[:len(data)]
        if data:
            d_file.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        d_text, d_offs, n = sp.SplitLinesDevice(d_file)
        lines = getline_split(data)
        want_text, want_offs = packed(lines)
        assert n == len(lines)
        assert d_text.cpu().numpy().tobytes() == want_text
        assert np.array_equal(d_offs.cpu().numpy().astype(np.uint64), want_offs)


@pytest.mark.gpu
def test_split_then_encode_gpu(oracle):
    """file image -> splitter -> EncodeDevice == Encode of the getline lines."""
    import torch
    from tests import fixtures
    from sentencepiece_amd.processor import SentencePieceProcessor
    sp = SentencePieceProcessor(model_proto=fixtures.model_blob("uni1k"), device=0)
    rng = random.Random(11)
    words = ["hello", "world", "こんにちは", "tokenizer", "a", "", "  spaced  "]
    lines = [" ".join(rng.choice(words) for _ in range(rng.randint(0, 30))) for _ in range(5000)]
    data = ("\n".join(lines) + "\n").encode()
    d_file = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda:0")[:len(data)]
    d_file.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    d_text, d_offs, n = sp.SplitLinesDevice(d_file)
    assert n == len(lines)
    ids, id_offs, total = sp.EncodeDevice(d_text, d_offs)
    ids, id_offs = ids[:total].cpu().numpy(), id_offs.cpu().numpy()
    o = oracle.load(fixtures.model_blob("uni1k"))
    want_text, want_offs = packed([x.encode() for x in lines])
    want_ids, want_io = o.encode_batch(np.frombuffer(want_text, dtype=np.uint8), want_offs)
    assert np.array_equal(id_offs.astype(np.int64), np.asarray(want_io).astype(np.int64))
    assert np.array_equal(ids, np.asarray(want_ids)[:total])
