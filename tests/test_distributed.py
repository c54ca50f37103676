"""N > 1 path on CPU: two gloo ranks render their shares (the CPU checker stands in for the kernel here),
the product's combine() sums the accumulators, rank 0 compares with the one-call image."""
import os
import socket

import numpy as np
import pytest

import raytrace_amd as rt
from raytrace_amd import distributed as rtd


def test_split_frame_covers_everything_once():
    for world, groups in ((1, 1), (2, 1), (8, 1), (8, 2), (8, 8), (4, 2)):
        shares = rtd.split_frame(1080, 4096, world, groups)
        cover = np.zeros((1080, 4096 // 64), dtype=np.int32)  # rows x sample blocks of 64
        for s in shares:
            assert s.s0 % 64 == 0 and s.s1 % 64 == 0
            cover[s.y0:s.y1, s.s0 // 64:s.s1 // 64] += 1
        assert (cover == 1).all(), (world, groups)
    with pytest.raises(ValueError):
        rtd.split_frame(100, 16, 3, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, row_groups, out_path):
    import torch
    import torch.distributed as dist
    import oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), 40, 24, 6, rng_mode=rt.RNG_COUNTER)
    share = rtd.split_frame(p.height, 6, world, row_groups)[rank]
    acc, _ = oracle.render(scene, rtd.apply_share(p, share))
    t = torch.from_numpy(acc)
    rtd.combine(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("row_groups", [1, 2])
def test_two_ranks_compose_the_frame(tmp_path, row_groups):
    import torch.multiprocessing as mp
    import oracle
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, _free_port(), row_groups, out), nprocs=2, join=True)
    got = np.load(out)
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), 40, 24, 6, rng_mode=rt.RNG_COUNTER)
    want, _ = oracle.render(scene, p)
    if row_groups == 2:   # disjoint pixels, full sample range each: exact
        assert np.array_equal(got, want)
    else:                 # partial sums over sample ranges: order of the f32 additions differs
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
