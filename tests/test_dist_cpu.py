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
