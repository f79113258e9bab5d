"""The N > 1 path on CPU: two `gloo` ranks shard 4 cascades 2 + 2 exactly as bench.py does on GPUs
(godotoceanwaves_amd/sharding.py), each rank computes its own cascades (with the CPU oracle standing in for the
device kernels -- this test is about the partitioning and the gather layout, not the arithmetic), all_gather,
and rank 0 compares the gathered arrays with a single-process run of all 4 cascades."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_owned_cascades_partition():
    from godotoceanwaves_amd import sharding
    for world, per in ((1, 4), (2, 2), (8, 1), (4, 2)):
        seen = []
        for r in range(world):
            own = sharding.owned_cascades(r, world, per)
            assert len(own) == per
            for l, g in enumerate(own):
                assert sharding.owner_of(g, per) == (r, l)
            seen += own
        assert seen == list(range(world * per))
    with pytest.raises(ValueError):
        sharding.owned_cascades(2, 2, 1)


def _worker(rank, world, port, n, per_rank, frames, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import helpers as H
    from godotoceanwaves_amd import sharding
    from godotoceanwaves_amd.presets import UPDATE_DELTA
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ids = sharding.owned_cascades(rank, world, per_rank)
    g = H.oracle_generator(n, ids)
    for _ in range(frames):
        g.update_all(UPDATE_DELTA)
    disp = torch.from_numpy(np.stack([g.displacement(i) for i in range(per_rank)]).view(np.uint8))
    norm = torch.from_numpy(np.stack([g.normal(i) for i in range(per_rank)]).view(np.uint8))
    gathered = sharding.alloc_gather_buffers(torch, world, disp, norm)
    sharding.gather_maps(dist, gathered, disp, norm)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "disp.npy"), gathered[0].numpy())
        np.save(os.path.join(out_dir, "norm.npy"), gathered[1].numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp
    import helpers as H
    from godotoceanwaves_amd.presets import UPDATE_DELTA
    n, per_rank, world, frames = 128, 2, 2, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n, per_rank, frames, str(tmp_path)), nprocs=world, join=True)
    disp = np.load(tmp_path / "disp.npy").view(np.uint16)
    norm = np.load(tmp_path / "norm.npy").view(np.uint16)
    assert disp.shape == (world, per_rank, n, n, 4)  # gathered as bytes, viewed back as RGBA16F bits
    g = H.oracle_generator(n, list(range(world * per_rank)))
    for _ in range(frames):
        g.update_all(UPDATE_DELTA)
    for c in range(world * per_rank):
        r, l = divmod(c, per_rank)
        assert np.array_equal(disp[r, l], g.displacement(c)), c
        assert np.array_equal(norm[r, l], g.normal(c)), c
