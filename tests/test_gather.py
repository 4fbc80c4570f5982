"""spmx_all_gather_ids (include/spmx.h, csrc/gather.cc): the ids of every rank on every rank, from a C host.

CPU legs: the world > 1 logic with ranks = threads of this process -- every rank encodes its shard through the PRODUCT's
encode under the emulator (tests/emulib.py), then all call spmx_all_gather_ids against tests/emu/libfake_rccl.so (a
stand-in for librccl over host memory, SPMX_RCCL_LIB); every rank's gathered CSR is compared with the oracle's encode
of the whole corpus.  -m gpu: libspmx.so against the real librccl at world 1 (one GPU per box): communicator set-up
through the spmx_rccl_* helpers, the gather, the rebased offsets."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from sentencepiece_amd import sharding
from tests import fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gather_rank(lib, comm, rank, world, ids, io, cap_ids, cap_offs):
    """-> (all_ids, all_offsets, rank_sentences, rank_ids) of one rank (arrays in 'device' = host memory)."""
    n = len(io) - 1
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    io = np.ascontiguousarray(io, dtype=np.uint64)
    all_ids = np.full(cap_ids, -7, dtype=np.int32)
    all_offs = np.full(cap_offs, 0xCDCD, dtype=np.uint64)
    scratch = np.zeros(int(lib.spmx_gather_scratch_words(world)), dtype=np.uint64)
    rs = np.zeros(world + 1, dtype=np.uint64)
    ri = np.zeros(world + 1, dtype=np.uint64)
    rc = lib.spmx_all_gather_ids(comm, rank, world, ids.ctypes.data, len(ids), io.ctypes.data, n, all_ids.ctypes.data, cap_ids,
                                 all_offs.ctypes.data, cap_offs, scratch.ctypes.data, rs.ctypes.data, ri.ctypes.data, None)
    return rc, all_ids, all_offs, rs, ri


@pytest.fixture(scope="module")
def emu_lib():
    os.environ["SPMX_RCCL_LIB"] = os.path.join(ROOT, "tests", "emu", "libfake_rccl.so")    # (read at the first gather call)
    from tests import emulib
    return emulib.EmuLib()


@pytest.mark.parametrize("world,cuts", [(2, None), (3, None), (3, [0, 0, 700]), (4, [0, 100, 100, 450])])
@pytest.mark.parametrize("model", ["uni32k", "bpe1k"])
def test_all_gather_ids_over_threads(model, world, cuts, emu_lib, oracle, corpora):
    blob = fixtures.model_blob(model)
    text, offs = fixtures.head(*corpora["synth20k"], 900)
    oids, oio = oracle.load(blob).encode_batch(text, offs)
    n = len(offs) - 1
    if cuts is None:
        sb = sharding.shard_bounds(offs, world)                         # byte-balanced contiguous shards
        bounds = [(int(sb[r]), int(sb[r + 1])) for r in range(world)]
    else:
        bounds = [(cuts[r], cuts[r + 1] if r + 1 < world else n) for r in range(world)]   # uneven, with empty shards
    lib = emu_lib.lib
    uid = (C.c_char * 128)()
    assert lib.spmx_rccl_unique_id(uid) == 0, lib.spmx_gather_last_error()
    handles = [emu_lib.load(blob, classes=None) for _ in range(world)]
    out = [None] * world

    def run(rank):
        comm = C.c_void_p()
        assert lib.spmx_rccl_comm_init(C.byref(comm), world, rank, uid) == 0
        a, b = bounds[rank]
        t = text[int(offs[a]):int(offs[b])]
        o = (offs[a:b + 1] - offs[a]).astype(np.uint64)
        ids, io = handles[rank].encode_batch(t, o) if b > a else (np.zeros(0, np.int32), np.zeros(1, np.uint64))
        out[rank] = _gather_rank(lib, comm, rank, world, ids, io, len(oids) + 5, n + 3)
        lib.spmx_rccl_comm_destroy(comm)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive(), "a rank hangs"
    for rank in range(world):
        rc, all_ids, all_offs, rs, ri = out[rank]
        assert rc == 0, lib.spmx_gather_last_error()
        np.testing.assert_array_equal(all_offs[:n + 1], np.asarray(oio))
        np.testing.assert_array_equal(all_ids[:len(oids)], np.asarray(oids))
        assert (all_ids[len(oids):] == -7).all() and (all_offs[n + 1:] == 0xCDCD).all()      # nothing written past the CSR
        assert rs.tolist() == [bounds[0][0]] + [b for _, b in bounds] and int(ri[-1]) == len(oids)


def test_all_gather_ids_reports_a_small_capacity(emu_lib):
    lib = emu_lib.lib
    uid = (C.c_char * 128)()
    assert lib.spmx_rccl_unique_id(uid) == 0
    comm = C.c_void_p()
    assert lib.spmx_rccl_comm_init(C.byref(comm), 1, 0, uid) == 0
    rc, _, _, _, _ = _gather_rank(lib, comm, 0, 1, np.arange(10, dtype=np.int32), np.array([0, 4, 10], dtype=np.uint64), 9, 3)
    assert rc == 8 and b"10 ids" in lib.spmx_gather_last_error()
    rc, all_ids, all_offs, _, _ = _gather_rank(lib, comm, 0, 1, np.arange(10, dtype=np.int32), np.array([0, 4, 10], dtype=np.uint64), 10, 3)
    assert rc == 0 and all_offs.tolist() == [0, 4, 10] and all_ids.tolist() == list(range(10))
    assert lib.spmx_all_gather_ids(comm, 2, 1, None, 0, None, 0, None, 0, None, 0, None, None, None, None) == 3
    lib.spmx_rccl_comm_destroy(comm)


def test_all_gather_ids_one_rank_too_small_every_rank_returns_8(emu_lib):
    """Ranks size their output buffers from what they know -- one of them too small: the capacities travel with the counts,
    so EVERY rank returns 8 and none posts a transfer its peer will never match (round-4 ADVICE: the small rank used to
    return alone, the others hung in ncclSend / ncclRecv).  Then all retry with enough room."""
    lib = emu_lib.lib
    world = 3
    uid = (C.c_char * 128)()
    assert lib.spmx_rccl_unique_id(uid) == 0
    shards = [(np.arange(5, dtype=np.int32), np.array([0, 2, 5], dtype=np.uint64)),
              (np.arange(5, 12, dtype=np.int32), np.array([0, 7], dtype=np.uint64)),
              (np.zeros(0, dtype=np.int32), np.array([0], dtype=np.uint64))]
    caps = [(12, 4), (11, 4), (12, 4)]          # rank 1's id buffer is one short of the job's 12 ids
    first, second = [None] * world, [None] * world

    def run(rank):
        comm = C.c_void_p()
        assert lib.spmx_rccl_comm_init(C.byref(comm), world, rank, uid) == 0
        first[rank] = _gather_rank(lib, comm, rank, world, shards[rank][0], shards[rank][1], *caps[rank])
        msg = lib.spmx_gather_last_error()
        first[rank] = first[rank] + (msg,)
        second[rank] = _gather_rank(lib, comm, rank, world, shards[rank][0], shards[rank][1], 12, 4)
        lib.spmx_rccl_comm_destroy(comm)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
        assert not t.is_alive(), "a rank hangs"
    for rank in range(world):
        assert first[rank][0] == 8 and b"rank 1" in first[rank][5], (rank, first[rank][0], first[rank][5])
        rc, all_ids, all_offs, rs, ri = second[rank]
        assert rc == 0 and all_ids.tolist() == list(range(12)) and all_offs.tolist() == [0, 2, 5, 12]


@pytest.mark.gpu
def test_all_gather_ids_on_the_gpu_world_1(oracle, corpora):
    """libspmx.so + the real librccl on the one GPU of the box: communicator through the spmx_rccl_* helpers, counts
    all-gather, self copy, rebase kernel; the product encode feeds it."""
    import torch
    from sentencepiece_amd import _capi
    from sentencepiece_amd.processor import SentencePieceProcessor
    lib = _capi.lib()
    blob = fixtures.model_blob("uni32k")
    text, offs = fixtures.head(*corpora["synth20k"], 5000)
    sp = SentencePieceProcessor(model_proto=blob)
    dev = torch.device("cuda", 0)
    d_ids, d_io, total = sp.EncodeDevice(torch.from_numpy(text).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev))
    n = len(offs) - 1
    uid = (C.c_char * 128)()
    assert lib.spmx_rccl_unique_id(uid) == 0, lib.spmx_gather_last_error()
    comm = C.c_void_p()
    assert lib.spmx_rccl_comm_init(C.byref(comm), 1, 0, uid) == 0, lib.spmx_gather_last_error()
    all_ids = torch.full((total + 8,), -7, dtype=torch.int32, device=dev)
    all_offs = torch.zeros(n + 2, dtype=torch.int64, device=dev)
    scratch = torch.zeros(int(lib.spmx_gather_scratch_words(1)), dtype=torch.int64, device=dev)
    rs = np.zeros(2, dtype=np.uint64)
    ri = np.zeros(2, dtype=np.uint64)
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.spmx_all_gather_ids(comm, 0, 1, d_ids.data_ptr(), total, d_io.data_ptr(), n, all_ids.data_ptr(), total + 8,
                                 all_offs.data_ptr(), n + 2, scratch.data_ptr(), rs.ctypes.data, ri.ctypes.data, stream)
    assert rc == 0, lib.spmx_gather_last_error()
    torch.cuda.synchronize()
    oids, oio = oracle.load(blob).encode_batch(text, offs)
    np.testing.assert_array_equal(all_offs[:n + 1].cpu().numpy().astype(np.uint64), np.asarray(oio))
    np.testing.assert_array_equal(all_ids[:total].cpu().numpy(), np.asarray(oids))
    assert int(all_ids[total]) == -7 and rs.tolist() == [0, n] and ri.tolist() == [0, total]
    assert lib.spmx_rccl_comm_destroy(comm) == 0


@pytest.mark.parametrize("world", [1, 3])
def test_cpp_host_gathers_over_threads(world, emu_lib):
    """tests/cpp/gather_test.cc: the C++ a host of the reference would write (facade Load / EncodeBatchDevice /
    AllGatherIds), ranks = threads, device emulated, RCCL stood in for."""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "gather_test.cc")
    out = os.path.join(ROOT, "tests", "cpp", "gather_test_emu")
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-o", out, src, "-L" + emu_dir, "-lspmx_emu",
                           "-Wl,-rpath," + emu_dir])
    env = dict(os.environ, SPMX_RCCL_LIB=os.path.join(emu_dir, "libfake_rccl.so"), SPMX_EMU_CUS="2")
    r = subprocess.run([out, os.path.join(fixtures.GOLDEN, "test_model.model"), os.path.join(fixtures.GOLDEN, "botchan.txt"), str(world)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok 600 "), (r.stdout, r.stderr)
