"""world_size-2 gloo test of the data-parallel driver (sharding rule + the single all_gather), on CPU with fake
compute functions — the GPU kernels are covered by the -m gpu tests."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stylesinger_amd import dist as ssd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, n_items=6):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = list(range(n_items))  # item i has length 10+i

    def infer_fn(mine):
        T = 16
        mel = torch.zeros(len(mine), T, 80)
        f0 = torch.zeros(len(mine), T)
        lens = torch.tensor([10 + i for i in mine], dtype=torch.int32)
        for j, i in enumerate(mine):
            mel[j, :10 + i] = float(i + 1)
            f0[j, :10 + i] = 100.0 + i
        return mel, f0, lens

    def vocode_fn(mel, f0, lens):
        return mel.mean(-1).repeat_interleave(4, dim=1)

    out = ssd.run_sharded(infer_fn, vocode_fn, items, rank, world, pad_T=16)
    q.put((rank, out["mel_all"][:, 0, 0].tolist(), out["lens_all"].tolist(), out["local_indices"], tuple(out["wav_local"].shape)))
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    assert ssd.shard_indices(7, 1, 3) == [1, 4]
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, first_vals, lens_all, local, wshape in res:
        # the gathered batch comes back in ITEM order on every rank
        assert first_vals == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
        assert lens_all == [10, 11, 12, 13, 14, 15]
        assert local == list(range(rank, 6, 2))
        assert wshape == (3, 64)


def test_two_rank_gloo_uneven_shards():
    """7 items over 2 ranks: rank 1 holds one item fewer; its shard is padded with an empty slot so that the single
    all_gather sees equal shapes (unequal B_local would hang the collective). Lengths travel bit-exact."""
    world, n = 2, 7
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ssd.shard_slots(n, world) == 4 and ssd.gathered_order(n, world) == [0, 2, 4, 6, 1, 3, 5, -1]
    for rank, first_vals, lens_all, local, wshape in res:
        assert first_vals == [float(i + 1) for i in range(n)]
        assert lens_all == [10 + i for i in range(n)]
        assert local == list(range(rank, n, 2))
        assert wshape == (len(local), 64)


def test_single_process_is_identity():
    mel, f0, lens = torch.randn(2, 5, 80), torch.randn(2, 5), torch.tensor([5, 3], dtype=torch.int32)
    a, b, c = ssd.gather_mels(mel, f0, lens)
    assert a is mel and b is f0 and c is lens


def _worker_forced(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    mel, f0 = torch.randn(3, 16, 80, generator=g), torch.rand(3, 16, generator=g) * 300
    lens = torch.tensor([16, 0, 9], dtype=torch.int32)
    plain = ssd.gather_mels(mel, f0, lens)                 # world of one: the short cut returns the inputs themselves
    short_cut = plain[0] is mel
    ssd.FORCE_COLLECTIVE = True                            # ... unless forced: the real all_gather_into_tensor branch (CPU form of it)
    try:
        m, f, l = ssd.gather_mels(mel, f0, lens)
        res = ssd.run_sharded(lambda items: (mel[:len(items)], f0[:len(items)], lens[:len(items)]), lambda a, b, c: a.sum(-1), [0, 1, 2], 0, 1, 16)
    finally:
        ssd.FORCE_COLLECTIVE = False
    q.put(dict(short_cut=short_cut, forced_is_copy=m is not mel, equal=bool(torch.equal(m, mel) and torch.equal(f, f0) and torch.equal(l, lens)),
               dtype=str(l.dtype), sharded=bool(torch.equal(res["mel_all"], mel) and torch.equal(res["lens_all"], lens))))
    dist.destroy_process_group()


def test_forced_collective_runs_the_all_gather_branch_in_a_world_of_one():
    """dist.FORCE_COLLECTIVE (env SS_FORCE_COLLECTIVE=1) lifts the world-size-1 short cut of gather_mels: how the 1-GPU box drives the RCCL
    branch (tests/test_gpu_round4.py). Here on CPU with gloo: payload packing, bit-cast lengths and gathered order come back exact."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_forced, args=(0, 1, _free_port(), q))
    p.start()
    out = q.get(timeout=120)
    p.join(timeout=60)
    assert out == dict(short_cut=True, forced_is_copy=True, equal=True, dtype="torch.int32", sharded=True), out


def _run_world(world, n_items):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_items)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def test_eight_rank_gloo_world_64_and_65_items():
    """BASELINE configs[2]'s world size (8 ranks, one per GPU of a node) on CPU: 64 items (8 per rank, the config as specified) and 65 (uneven: rank 0
    holds 9 items, the others 8 + one empty slot so that the single all_gather sees equal shapes). Every rank gets the whole mel batch back in
    ITEM order with the lengths bit-exact, and keeps exactly its own shard `x[rank::8]` (tasks/tts/tts_base.py:129-132) for vocoding.
    No 8-GPU node has been available to any round's driver: this and the one-device emulation of tests/test_gpu_round6.py are what covers N = 8."""
    world = 8
    for n in (64, 65):
        slots = ssd.shard_slots(n, world)
        assert slots == (8 if n == 64 else 9)
        order = ssd.gathered_order(n, world)
        assert sorted(g for g in order if g >= 0) == list(range(n)) and order.count(-1) == world * slots - n
        for rank, first_vals, lens_all, local, wshape in _run_world(world, n):
            assert first_vals == [float(i + 1) for i in range(n)], (n, rank)
            assert lens_all == [10 + i for i in range(n)]
            assert local == list(range(rank, n, world))
            assert wshape == (len(local), 64)
