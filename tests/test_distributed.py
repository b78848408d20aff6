"""Multi-process path on CPU (gloo, world size 2): ray sharding and the framebuffer reduction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import distributed as rd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, sppx = 96, 80, 5
    rs = np.random.RandomState(123)
    samples = torch.from_numpy(rs.rand(sppx, h, w, 3).astype(np.float32))   # identical on every rank
    fb = rd.Framebuffer(w, h)
    if mode == "samples":
        for s in rd.shard_samples(sppx, world, rank):
            fb.add(samples[s])
    else:
        for (x0, y0, x1, y1) in rd.shard_tiles(w, h, world, rank, tile=32):
            mask = torch.zeros((h, w), dtype=torch.bool)
            mask[y0:y1, x0:x1] = True
            for s in range(sppx):
                fb.add(samples[s], mask)
    fb.reduce()
    np.save(os.path.join(out_dir, "fb_%s_%d.npy" % (mode, rank)), fb.color.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_framebuffer_reduce_matches_single_process(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    w, h, sppx = 96, 80, 5
    rs = np.random.RandomState(123)
    samples = torch.from_numpy(rs.rand(sppx, h, w, 3).astype(np.float32))
    ref = rd.Framebuffer(w, h)
    for s in range(sppx):                      # what one process running the reference's loop accumulates
        ref.add(samples[s])
    for rank in range(world):
        got = np.load(tmp_path / ("fb_%s_%d.npy" % (mode, rank)))
        assert (got[..., 3] == sppx).all()
        np.testing.assert_allclose(got[..., 0:3], ref.color.numpy()[..., 0:3], rtol=2e-6, atol=1e-7)


def test_sharding_covers_everything_once():
    for world in (1, 2, 3, 8):
        seen = sorted(s for r in range(world) for s in rd.shard_samples(64, world, r))
        assert seen == list(range(64))
        cover = np.zeros((1080, 1920), np.int32)
        for r in range(world):
            for (x0, y0, x1, y1) in rd.shard_tiles(1920, 1080, world, r):
                cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all()


def test_running_mean_matches_reference_formula():
    # framebuffer::add: c = (c.rgb * c.w + acc) / (c.w + n), c.w += n  (libgi/framebuffer.cpp:14-17)
    fb = rd.Framebuffer(4, 3)
    a = torch.full((3, 4, 3), 0.25)
    b = torch.full((3, 4, 3), 1.0)
    fb.add(a)
    fb.add(b)
    assert torch.allclose(fb.color[..., 0:3], torch.full((3, 4, 3), 0.625)) and (fb.color[..., 3] == 2).all()
    img = fb.to_rgb8()
    assert img.shape == (3, 4, 3) and img[0, 0, 0] == int((0.625 ** (1 / 2.2)) * 255)


def _bench_worker(rank, world, port, out_dir):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    elapsed, rays = bench.whole_job(1.0 + rank, 100 * (rank + 1), "cpu", True)
    np.save(os.path.join(out_dir, "job_%d.npy" % rank), np.array([elapsed, rays]))
    dist.destroy_process_group()


def test_bench_reports_slowest_rank_and_all_rays(tmp_path):
    # bench.py: the job's time is the maximum over the ranks, its rays their sum, on every rank
    world = 2
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert np.load(tmp_path / ("job_%d.npy" % rank)).tolist() == [2.0, 300.0]
    import bench
    assert bench.whole_job(0.5, 7, "cpu", False) == (0.5, 7.0)


@pytest.mark.parametrize("w,h,tile,world", [(96, 64, 32, 2), (100, 37, 16, 3), (1920, 1080, 64, 8), (64, 64, 64, 4), (5, 3, 0, 2)])
def test_the_frame_library_deals_tiles_as_the_gloo_tests_assume(w, h, tile, world):
    """The C++ frame level (what runs on N GPUs) and rd.shard_tiles (what the world-size-2 gloo tests exercise)
    deal the same pixels to the same rank in the same order, and the ranks' shares partition the frame."""
    from raytracing_amd import frame
    seen = np.zeros(w * h, np.int32)
    for rank in range(world):
        pixels = frame.frame_share("tiles", w, h, tile, 1, world, rank)
        want = [y * w + x for (x0, y0, x1, y1) in rd.shard_tiles(w, h, world, rank, tile or 64)
                for y in range(y0, y1) for x in range(x0, x1)]
        assert pixels.tolist() == want
        seen[pixels] += 1
    assert (seen == 1).all()


@pytest.mark.parametrize("sppx,world", [(64, 8), (7, 2), (1, 4), (0, 2)])
def test_the_frame_library_deals_samples_as_the_gloo_tests_assume(sppx, world):
    from raytracing_amd import frame
    dealt = []
    for rank in range(world):
        samples = frame.frame_share("samples", 16, 16, 0, sppx, world, rank)
        assert samples.tolist() == rd.shard_samples(sppx, world, rank)
        dealt += samples.tolist()
    assert sorted(dealt) == list(range(sppx))


def test_frame_share_refuses_ranks_outside_the_world():
    from raytracing_amd import frame
    with pytest.raises(ValueError):
        frame.frame_share("tiles", 8, 8, 4, 1, 2, 2)
