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


def _gather_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    import oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), 40, 52, 4, rng_mode=rt.RNG_COUNTER)
    shares = rtd.split_interleaved(p.height, 4, world)  # 52 rows: six and a half strips over three ranks
    acc, _ = oracle.render(scene, rtd.apply_share(p, shares[rank]))
    t = torch.from_numpy(acc)
    for _ in range(2):  # the cached buffers serve a second frame
        rtd.gather_rows(t, shares, rank, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.destroy_process_group()


def test_three_ranks_gather_their_strips(tmp_path):
    """Disjoint shares: each rank sends the root its own rows only; the frame is the one-call frame bit for bit."""
    import torch.multiprocessing as mp
    import oracle
    out = str(tmp_path / "frame.npy")
    mp.spawn(_gather_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    got = np.load(out)
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), 40, 52, 4, rng_mode=rt.RNG_COUNTER)
    want, _ = oracle.render(scene, p)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert rtd.share_rows(rtd.split_interleaved(52, 4, 3)[2]) == list(range(16, 24)) + list(range(40, 48))
    with pytest.raises(ValueError):
        import torch
        rtd.gather_rows(torch.zeros((8, 4, 3)), rtd.split_frame(8, 4, 2), 0)  # sample ranges must be summed


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


def _gpu_worker(rank, world, port, mode, out_path):
    """One rank of the HIP path: both ranks share device 0 (gloo carries the sum; RCCL needs a device per rank)."""
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)  # a rank that hangs ends itself and says where
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    W, H, S = 200, 120, 8
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), W, H, S, rng_mode=rt.RNG_COUNTER)
    r = rt.Renderer(0)
    r.upload(scene, rt.ACCEL_LBVH)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    r.set_accumulator(accum.data_ptr(), W, H)
    shares = rtd.split_interleaved(H, S, world) if mode in ("strips", "gather") else rtd.split_frame(H, S, world)
    for frame in range(2):  # twice: the second frame's clear must not overtake the first frame's sums
        r.clear()
        if mode == "gather":
            r.render(rtd.apply_share(p, shares[rank]))
            rtd.gather_rows(accum, shares, rank, dst=0, after=r)
        else:
            rtd.render_and_combine(r, rtd.apply_share(p, shares[rank]), accum, dst=0, bands=3)
    torch.cuda.synchronize()
    if rank == 0:
        got = accum.cpu().numpy().copy()
        r.set_accumulator(0, 0, 0)
        r.clear()
        r.render(p)
        want, _ = r.readback(rgb8=False)
        np.save(out_path, np.stack([got / np.float32(S), want]))
    dist.barrier()
    r.close()
    dist.destroy_process_group()


def _rccl_single_worker(rank, world, port, out_path):
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    W, H, S = 320, 200, 64
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), W, H, S, rng_mode=rt.RNG_COUNTER)
    r = rt.Renderer(0)
    r.upload(scene, rt.ACCEL_LBVH)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    r.set_accumulator(accum.data_ptr(), W, H)
    frames = []
    for bands in (1, 3, 3):
        r.clear()
        rtd.render_and_combine(r, p, accum, dst=0, bands=bands)
        frames.append(accum.clone())  # on torch's stream, which render_and_combine ordered behind the sums: no host wait
    for _ in range(2):  # the gather of disjoint rows, through RCCL's send / receive, ordered the same way
        r.clear()
        r.render(p)
        rtd.gather_rows(accum, rtd.split_interleaved(H, S, 1), 0, dst=0, after=r)
        frames.append(accum.clone())
    torch.cuda.synchronize()
    r.set_accumulator(0, 0, 0)
    r.clear()
    r.render(p)
    want, _ = r.readback(rgb8=False)
    np.save(out_path, np.stack([f.cpu().numpy() / np.float32(S) for f in frames] + [want]))
    r.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_device_ordered_sum_with_rccl_on_one_rank(tmp_path):
    """The RCCL path of render_and_combine (collective queued behind the render by an event, next frame's clear behind
    the collective) with a process group of one rank: nothing waits on the host between render, sum and the copy
    that reads the result, and the frame must still be the one-call frame."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "frames.npy")
    mp.spawn(_rccl_single_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    frames = np.load(out)
    for f in frames[:-1]:
        assert np.array_equal(f.view(np.uint32), frames[-1].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strips", "gather", "samples"])
def test_two_ranks_on_one_gpu_hip_path(tmp_path, mode):
    """The N > 1 flow with the HIP kernels: shares rendered band by band, each band's sum queued behind its render
    on the device.  Strips: disjoint pixels, exact.  Sample ranges: the two partial sums added once, float32 rounding."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "frames.npy")
    mp.spawn(_gpu_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    got, want = np.load(out)
    assert want.mean() > 0.05
    if mode in ("strips", "gather"):
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
