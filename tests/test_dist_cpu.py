"""world_size-2 gloo rehearsal of the data-parallel path: static sharding + ordered gather."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from marie_icr_amd.dist import gather_in_order, shard_indices


def test_shard_indices_partition():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in shard_indices(n, r, world))
            assert seen == list(range(n))
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def _worker(rank, world, port, n_items, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_items, rank, world)
    local = [{"page": i, "text": f"PAGE-{i}", "rank": rank} for i in mine]
    out = gather_in_order(local, n_items, dist)
    q.put((rank, [o["page"] for o in out], [o["rank"] for o in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_restores_page_order():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_items = 7  # ragged: rank 0 gets 4 pages, rank 1 gets 3
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, pages, ranks in got:
        assert pages == list(range(n_items))
        assert ranks == [i % 2 for i in range(n_items)]


class _HostCtx:
    """stands in for marie_icr_amd._lib.Context in the gloo rehearsal: "device" memory is host memory"""

    def memcpy_dev(self, dst, src, n):
        import ctypes

        ctypes.memmove(dst, src, n)

    def synchronize(self):
        pass


class _TwoArenaModel:
    """like DitModel / TrocrModel: two packed weight arenas"""

    def __init__(self, fill):
        import numpy as np

        self.a = [np.full(1 << 16, fill, np.uint8), np.full(12345, fill, np.uint8)]

    def arenas(self):
        return [(x.ctypes.data, x.nbytes) for x in self.a]


def _bcast_worker(rank, world, port, q):
    import numpy as np
    import torch.distributed as dist

    from marie_icr_amd.dist import broadcast_arenas, shard_indices

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _TwoArenaModel(0)
    if rank == 0:                       # rank 0 "packed the weights"
        rng = np.random.default_rng(7)
        for x in m.a:
            x[:] = rng.integers(0, 256, x.shape, dtype=np.uint8)
    broadcast_arenas(m, _HostCtx(), dist, src=0, device="cpu")
    pages = shard_indices(9, rank, world)      # bench.py's page partition of a 9-page job
    q.put((rank, [int(x.sum()) for x in m.a], pages))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_weight_arena_broadcast():
    """the start-up collective of bench.py --workload dit_trocr (RCCL there, gloo here)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (sums, pages)) for r, sums, pages in [q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] and got[0][0][0] > 0
    assert sorted(got[0][1] + got[1][1]) == list(range(9))
