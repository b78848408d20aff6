"""rtgi_hip_closest_hit_device_noalloc (the entry point that allocates nothing) against rtgi_hip_closest_hit_device,
through the C ABI: bit-identical hit records on two calls in a row, the second with a smaller input, for a scene
whose batches are sorted before tracing, and for a tree deeper than the stack levels kept in LDS."""
import numpy as np
import pytest

from raytracing_amd import scenes, tracer
from tests.test_gpu_parity import _chain_scene, _mixed_rays
from tests.util import assert_hits_equal

pytestmark = pytest.mark.gpu


def _both(t, rays):
    import torch
    d_rays = torch.from_numpy(rays).cuda()
    d_want = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
    d_got = torch.full((len(rays), 4), -1.0, dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_rays, d_want)
    t.closest_hit_device_noalloc(d_rays, d_got)
    torch.cuda.synchronize()
    want = d_want.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)
    got = d_got.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)
    assert_hits_equal(got, want, "noalloc vs closest_hit_device, %d rays" % len(rays))
    return want


def test_noalloc_two_calls_sorted_batches():
    pos, tri, _ = scenes.soup(60000, seed=3)
    t = tracer.HipBvhTracer(layout="indexed")
    assert t.interprete("hip", "reorder on")  # batches of 65536 rays and more go through the sort
    t.build(pos, tri)
    first = _both(t, _mixed_rays(pos, 300000, seed=21))
    second = _both(t, _mixed_rays(pos, 70001, seed=22))   # smaller: the array still holds the first call's contents
    third = _both(t, _mixed_rays(pos, 1000, seed=23))     # below the sort threshold
    assert (first["t"] != tracer.FLT_MAX).any() and (second["t"] != tracer.FLT_MAX).any() and len(third) == 1000
    # the largest batch the entry point takes is still sorted inside the fixed array (the sort's own scratch fits)
    import torch
    before = t.launch_stats()
    n = 1 << 21
    d_rays = torch.from_numpy(_mixed_rays(pos, n, seed=24)).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device_noalloc(d_rays, d_hits)
    torch.cuda.synchronize()
    after = t.launch_stats()
    assert after["launches"] == before["launches"] + 1 and after["sorted"] == before["sorted"] + 1
    t.close()


def test_noalloc_deep_tree_and_limit():
    import torch
    pos, tri, _ = _chain_scene(60)
    t = tracer.HipBvhTracer(layout="indexed")
    assert t.interprete("bvh", "sm")
    t.build(pos, tri)
    assert t.depth > 24  # the stack continues in the spill area
    _both(t, _mixed_rays(pos, 90000, seed=31))
    _both(t, _mixed_rays(pos, 5000, seed=32))
    # more rays than the fixed scratch is laid out for: refused, nothing launched
    n = (1 << 21) + 1
    d_rays = torch.zeros((n, 8), dtype=torch.float32, device="cuda")
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(tracer.RtgiHipError, match="at most"):
        t.closest_hit_device_noalloc(d_rays, d_hits)
    t.close()
