"""The N > 1 path on CPU: two gloo ranks shard one packed batch by bytes, encode
their shard (the oracle stands in for the GPU encode -- allowed in tests), and
all-gather ids + per-sentence offsets; every rank must end up with exactly the
single-process result in the original order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import fixtures


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model, corpus_name, n_sent, out_dir, engine="oracle"):
    import torch.distributed as dist
    from sentencepiece_amd import sharding
    from tests import oraclelib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        text, offs = fixtures.head(*fixtures.Corpora()[corpus_name], n_sent)
        if engine == "emu":      # the PRODUCT's encode (api.cc + kernels under the emulator), not the oracle
            from tests import emulib
            o = emulib.EmuLib().load(fixtures.model_blob(model), classes=None)
        else:
            o = oraclelib.OracleLib().load(fixtures.model_blob(model))

        def encode_fn(t, of):
            ids, io = o.encode_batch(t.numpy(), of.numpy().astype(np.uint64))
            return torch.from_numpy(np.asarray(ids)), torch.from_numpy(np.asarray(io).astype(np.int64)), len(ids)

        ids, io = sharding.encode_sharded(encode_fn, torch.from_numpy(text.copy()), offs, dist, torch.device("cpu"))
        np.save(os.path.join(out_dir, "ids%d.npy" % rank), ids.numpy())
        np.save(os.path.join(out_dir, "io%d.npy" % rank), io.numpy())
        # steady-state gatherer: capacities agreed once (reserve), then batches of different size -- an empty one among
        # them -- through one IdGatherer, two gathers in flight
        g = sharding.IdGatherer(dist, torch.device("cpu"), wire_dtype=torch.int16, depth=2,
                                algo="p2p" if world == 4 else "all_gather")
        g.reserve(16, 2)
        sizes = lambda r, step: (3 + r, 9 - 2 * r if 9 - 2 * r > 0 else 0, 0 if r == 1 else 5)[step]      # noqa: E731
        for step in range(3):
            k = sizes(rank, step)
            part = torch.arange(k, dtype=torch.int32) + 100 * rank
            g(part, k, torch.tensor([0, k], dtype=torch.int64))
            got, goffs = g.result()
            for r in range(world):
                kk = sizes(r, step)
                assert got[r].tolist() == [100 * r + i for i in range(kk)] and got[r].dtype == torch.int32
                assert goffs[r].tolist() == [0, kk]
        g.wait()
        # exact sizes point to point: every rank sends its `total` ids and its per-sentence counts (int32) to every
        # peer, nothing is padded to the largest rank; sentences per rank differ too, and one rank sends nothing
        gx = sharding.IdGatherer(dist, torch.device("cpu"), wire_dtype=torch.int16, depth=2, algo="p2p_exact")
        for step in range(3):
            k = sizes(rank, step)
            nsent = 0 if k == 0 else 1 + (rank + step) % 3
            cuts = np.linspace(0, k, nsent + 1).astype(np.int64)
            part = torch.arange(k, dtype=torch.int32) + 100 * rank
            # the counts travel in one, two or four bytes by the SENDER's bound on ids per sentence (ranks differ on purpose)
            gx(part, k, torch.from_numpy(cuts), max_count=(None, 200, 40000)[(rank + step) % 3])
            got, goffs = gx.result()
            for r in range(world):
                kk = sizes(r, step)
                ns = 0 if kk == 0 else 1 + (r + step) % 3
                assert got[r].tolist() == [100 * r + i for i in range(kk)] and got[r].dtype == torch.int32
                assert goffs[r].tolist() == np.linspace(0, kk, ns + 1).astype(np.int64).tolist()
        gx.wait()
        # counts beyond a byte / beyond 32767 keep their value through the two-byte form
        gy = sharding.IdGatherer(dist, torch.device("cpu"), wire_dtype=torch.int16, depth=3, algo="p2p_exact")
        big = 40000 + rank
        gy(torch.zeros(big + 3, dtype=torch.int32), big + 3, torch.tensor([0, big, big + 3], dtype=torch.int64), max_count=65535)
        got, goffs = gy.result()
        for r in range(world):
            assert goffs[r].tolist() == [0, 40000 + r, 40003 + r] and got[r].numel() == 40003 + r
        gy.wait()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,corpus,n,world", [("test_model", "botchan", 700, 2), ("bpe1k", "edge", 66, 2),
                                                   ("test_model", "botchan", 301, 4), ("bpe1k", "edge", 3, 4),
                                                   ("test_model", "botchan", 500, 8)])
def test_gloo_ranks(model, corpus, n, world, tmp_path, oracle, corpora):
    """world 2, 4 and 8; uneven shards (301 sentences over 4 ranks by bytes) and empty ones (3 sentences over 4 ranks)."""
    mp.spawn(_worker, args=(world, _free_port(), model, corpus, n, str(tmp_path)), nprocs=world, join=True)
    text, offs = fixtures.head(*corpora[corpus], n)
    ids, io = oracle.load(fixtures.model_blob(model)).encode_batch(text, offs)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / ("ids%d.npy" % r)), ids)
        np.testing.assert_array_equal(np.load(tmp_path / ("io%d.npy" % r)), io.astype(np.int64))


@pytest.mark.parametrize("model,corpus,n,world", [("uni32k", "synth20k", 900, 2), ("bpe32k", "synth20k", 500, 3)])
def test_gloo_ranks_drive_the_product_encode(model, corpus, n, world, tmp_path, oracle, corpora):
    """The same N > 1 path with the product's own encode on every rank (csrc/api.cc and the kernels under the CPU
    emulator: classify with the plain scan, word rounds, general launches) instead of the oracle; the gathered CSR of
    every rank equals the oracle's encode of the whole batch."""
    from tests import emulib
    emulib.lib()            # (built once here, not by every rank at the same time)
    mp.spawn(_worker, args=(world, _free_port(), model, corpus, n, str(tmp_path), "emu"), nprocs=world, join=True)
    text, offs = fixtures.head(*corpora[corpus], n)
    ids, io = oracle.load(fixtures.model_blob(model)).encode_batch(text, offs)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / ("ids%d.npy" % r)), ids)
        np.testing.assert_array_equal(np.load(tmp_path / ("io%d.npy" % r)), io.astype(np.int64))


def test_shard_bounds_balance():
    from sentencepiece_amd import sharding
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 500, size=10000)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for world in (1, 2, 3, 8):
        b = sharding.shard_bounds(offs, world)
        assert b[0] == 0 and b[-1] == len(lens) and (np.diff(b) >= 0).all()
        per = [int(offs[b[r + 1]] - offs[b[r]]) for r in range(world)]
        assert max(per) - min(per) <= 1000
    # degenerate: fewer sentences than ranks, empty batch
    # (the one sentence goes to the rank whose byte target it reaches first; the others get empty shards)
    b = sharding.shard_bounds(np.array([0, 5], dtype=np.uint64), 4)
    assert b.tolist() == [0, 1, 1, 1, 1]
    b = sharding.shard_bounds(np.array([0, 5, 5, 9], dtype=np.uint64), 8)
    assert b[0] == 0 and b[-1] == 3 and (np.diff(b) >= 0).all() and np.diff(b).sum() == 3
    assert sharding.shard_bounds(np.array([0], dtype=np.uint64), 2).tolist() == [0, 0, 0]
