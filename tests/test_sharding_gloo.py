"""N>1 path on CPU: world_size-2 gloo ranks shard baselines with the reference's
block rule, generate only their own block, and agree on a max-over-ranks time."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def test_split_counts_matches_reference_rule():
    from hydra_pspec_amd.sharding import split_counts, split_data_for_scatter, block_range
    assert split_counts(10, 4) == [3, 3, 2, 2]
    assert split_counts(8192, 8) == [1024] * 8
    assert split_counts(5, 5) == [1] * 5
    assert split_data_for_scatter(list(range(7)), 3) == [[0, 1, 2], [3, 4], [5, 6]]
    assert block_range(10, 4, 2) == (6, 8)
    with pytest.raises(ValueError):
        split_counts(3, 4)


def _worker(rank, world, port, q):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hydra_pspec_amd import synthetic
    from hydra_pspec_amd.sharding import block_range
    lo, hi = block_range(5, world, rank)
    d = synthetic.make_baselines(16, 4, 2, k0=lo, nbl=hi - lo, dense=False)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # checksum of the shard, gathered only for the test (the data path has no collective)
    cs = [None] * world
    dist.all_gather_object(cs, (lo, hi, float(np.abs(d["vis"]).sum())))
    if rank == 0:
        q.put((float(t.item()), cs))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, cs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == pytest.approx(0.2)
    assert [(c[0], c[1]) for c in cs] == [(0, 3), (3, 5)]
    sys.path.insert(0, str(REPO))
    from hydra_pspec_amd import synthetic
    full = synthetic.make_baselines(16, 4, 2, k0=0, nbl=5, dense=False)
    assert cs[0][2] == pytest.approx(float(np.abs(full["vis"][:3]).sum()), rel=1e-14)
    assert cs[1][2] == pytest.approx(float(np.abs(full["vis"][3:]).sum()), rel=1e-14)
