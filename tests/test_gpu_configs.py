"""The benchmark configurations at their own sizes, through the C ABI on a real MI355X, against the CPU oracle:

1. the plumbing frame: render-data/tri.obj and 2tri.obj with the cameras of scripts/a1-tri and a2-tris
   (at 0 0 0, look 0 0 -1, up 0 1 0), 640x480, pixel centres, primary closest_hit on every pixel;
2. Cornell-class scene: primary closest_hit + one shadow any_hit per hit (in test_gpu_wavefront.py and
   test_gpu_shading.py);
3. Sponza-class scene of ~262 k triangles, 1280x720, one diffuse bounce per primary hit: closest_hit on the
   incoherent bounce batch;
4. the 2^20-triangle soup at 1920x1080 with its bounce batch (the workload bench.py times);
5. multi-million-triangle deep trees with mixed closest_hit / any_hit batches: a skewed scene whose tree is deep but
   whose rays keep few nodes pending (test_gpu_parity.py), and below a scene built so that most rays hold more
   pending nodes than the LDS part of the stack has levels (and than the reference's fixed 25 slots).

Bar: hit records bit-identical to the oracle's on a sample of at least 1/16 of the batch (all of it where the
oracle finishes in seconds), plus properties that hold for every ray."""
import numpy as np
import pytest

from oracle import oracle as orc
from raytracing_amd import scenes, tracer, wavefront
from tests.util import FLT_MAX, assert_hits_equal, obj_scene

pytestmark = pytest.mark.gpu


def _hits_of(d_hits):
    return d_hits.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)


@pytest.mark.parametrize("name,layout", [("tri.obj", "flat"), ("2tri.obj", "flat"), ("2tri.obj", "indexed")])
def test_config1_plumbing_frame(name, layout):
    import torch
    pos, tri, _ = obj_scene(name)
    t = tracer.HipBvhTracer(layout=layout)
    t.build(pos, tri)
    cam = ([0, 0, -1], [0, 0, 0], [0, 1, 0], 65.0, 640, 480)
    d_rays = t.camera_rays_device(*cam)  # jitter off: pixel centres
    rays = d_rays.cpu().numpy()
    assert rays.tobytes() == orc.cam_rays(*cam).tobytes()
    d_hits = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_rays, d_hits)
    hits = _hits_of(d_hits)
    o = orc.Bbvh(pos, tri, layout=layout)
    assert_hits_equal(hits, o.closest_hit(rays), "%s %s frame" % (name, layout))
    seq = orc.Seq(pos, tri).closest_hit(rays)  # brute force sees the same picture
    assert ((hits["t"] != FLT_MAX) == (seq["t"] != FLT_MAX)).all()
    assert 0.005 < (hits["t"] != FLT_MAX).mean() < 0.9
    assert_hits_equal(t.closest_hit(rays), hits, "host-buffer path")


def _bounce_batch(t, cam, sample):
    """Jittered camera rays, closest_hit, one diffuse bounce off every hit: what bench.py sets up."""
    import torch
    w, h = cam[4], cam[5]
    d_off = t.pixel_offsets_device(w * h, sample, seed=0)
    d_primary = t.camera_rays_device(*cam, d_offsets=d_off)
    d_hits = torch.empty((w * h, 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_primary, d_hits)
    d_bounce, _ = t.bounce_rays_device(d_primary, d_hits, seed=sample, bounce=1)
    torch.cuda.synchronize()
    return d_primary, d_hits, d_bounce.contiguous()


def _check_bounce_batch(t, o, pos, tri_reported, d_bounce, stride, what):
    """closest_hit and any_hit on a bounce batch: oracle on every stride-th ray (records and per-ray counts),
    properties on all of them."""
    import torch
    n = d_bounce.shape[0]
    rays = d_bounce.cpu().numpy()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    d_cnt = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    t.closest_hit_device(d_bounce, d_hits, d_cnt)
    t.any_hit_device(d_bounce, d_occ)
    torch.cuda.synchronize()
    hits, occ, cnt = _hits_of(d_hits), d_occ.cpu().numpy(), d_cnt.cpu().numpy().astype(np.uint32)
    sample = np.arange(0, n, stride)
    want, want_counts = o.closest_hit(rays[sample], counts=True)
    assert_hits_equal(hits[sample], want, what)
    assert (cnt[sample] == want_counts[:, 0:2]).all()
    assert (occ[sample] == o.any_hit(rays[sample])).all()
    valid = hits["t"] != FLT_MAX
    assert (occ == valid).all()                                    # any_hit <=> closest_hit finds something
    assert (hits["beta"][valid] > 0).all() and (hits["gamma"][valid] > 0).all()
    assert (hits["beta"][valid] + hits["gamma"][valid] <= 1).all()
    assert (hits["ref"][valid] < len(tri_reported)).all()
    short = rays[valid].copy()                                     # nothing lies in front of a closest hit
    short[:, 7] = hits["t"][valid]
    d_short = torch.from_numpy(short).cuda()
    d_hits2 = torch.empty((len(short), 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_short, d_hits2)
    assert (_hits_of(d_hits2)["t"] == FLT_MAX).all()
    v = pos[tri_reported[hits["ref"][valid]]].astype(np.float64)   # the barycentric point lies on the ray
    b, g = hits["beta"][valid, None].astype(np.float64), hits["gamma"][valid, None].astype(np.float64)
    p_tri = v[:, 0] * (1 - b - g) + v[:, 1] * b + v[:, 2] * g
    p_ray = rays[valid, 0:3] + hits["t"][valid, None].astype(np.float64) * rays[valid, 3:6]
    scale = max(1.0, float(np.abs(pos).max()))
    assert np.abs(p_tri - p_ray).max() < 2e-4 * scale
    # the sorted launch and the launch in the given order agree on every ray
    assert t.interprete("hip", "reorder off")
    d_hits3 = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_bounce, d_hits3)
    assert t.interprete("hip", "reorder on")
    d_hits4 = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_bounce, d_hits4)
    assert t.interprete("hip", "reorder auto")
    torch.cuda.synchronize()
    assert torch.equal(d_hits3, d_hits) and torch.equal(d_hits4, d_hits)
    return hits, cnt


def test_config3_sponza_class_260k_bounce_rays():
    pos, tri, _ = scenes.sponza_class(262144, seed=0)
    assert abs(len(tri) - 262144) <= 0.05 * 262144
    t = tracer.HipBvhTracer(layout="flat")
    t.build(pos, tri)
    o = orc.Bbvh(pos, tri, layout="flat")
    assert t.nodes().tobytes() == o.nodes().tobytes()
    tri_reported = tri[t.triangle_order()]                         # flat layout: refs index the permuted list
    cam = (np.array([1, 0, 0], np.float32), [-516, 300, 0], [0, 1, 0], 65.0, 1280, 720)  # scripts/a2-sponza
    d_primary, d_hits, d_bounce = _bounce_batch(t, cam, sample=0)
    primary, phits = d_primary.cpu().numpy(), _hits_of(d_hits)
    every = np.arange(0, len(primary), 16)
    assert_hits_equal(phits[every], o.closest_hit(primary[every]), "sponza-class primary")
    assert (phits["t"] != FLT_MAX).mean() > 0.95                   # closed room
    # the bounce rays are the host definition's (same hits, same counter-based random numbers)
    want_bounce, _ = wavefront.bounce_rays(pos, tri_reported, primary, phits, seed=0, bounce=1)
    got_bounce = d_bounce.cpu().numpy()
    assert got_bounce.shape == want_bounce.shape
    assert got_bounce[:, 0:3].tobytes() == want_bounce[:, 0:3].tobytes()   # origins: same fp32 operations
    np.testing.assert_allclose(got_bounce[:, 3:6], want_bounce[:, 3:6], rtol=0, atol=5e-5)
    assert d_bounce.shape[0] > 800000
    _check_bounce_batch(t, o, pos, tri_reported, d_bounce, 16, "sponza-class bounce")
    t.close()


def test_config4_soup_1m_bounce_batch_1080p():
    pos, tri, _ = scenes.soup(1 << 20, seed=1)
    t = tracer.HipBvhTracer(layout="flat")
    t.build(pos, tri)
    o = orc.Bbvh(pos, tri, layout="flat")
    tri_reported = tri[t.triangle_order()]
    cam = (np.array([0, 0, -1], np.float32), [0, 0, 0], [0, 1, 0], 65.0, 1920, 1080)
    _, _, d_bounce = _bounce_batch(t, cam, sample=0)
    assert d_bounce.shape[0] > 1300000
    # (every ray against the CPU tracer: hit records, any-hit bytes and per-ray pop / test counts, 1.45 M of each)
    hits, cnt = _check_bounce_batch(t, o, pos, tri_reported, d_bounce, 1, "soup 1M bounce")
    assert 150 < cnt[:, 0].mean() < 250                            # incoherent rays: ~200 inner nodes each
    # the entry point that allocates nothing gives the same records
    import torch
    d_hits = torch.empty((d_bounce.shape[0], 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device_noalloc(d_bounce, d_hits)
    torch.cuda.synchronize()
    assert_hits_equal(_hits_of(d_hits), hits, "noalloc, soup 1M bounce")
    t.close()


def test_config5_deep_stack_scene_spills_at_4m_triangles():
    """2^22 triangles in 40 sheets of doubling size under spatial-median splits: a chain of 40 levels with a full
    subtree at each. Rays along the chain keep one node pending per sheet ahead, so most of them push beyond the
    levels of the stack that live in LDS. closest_hit and any_hit batches share launches; the oracle (growable stack)
    checks a sixteenth of the rays including per-ray counts, properties cover all of them."""
    import torch
    pos, tri, _ = scenes.stadium_sheets(1 << 22, sheets=40)
    assert len(tri) > 4000000
    t = tracer.HipBvhTracer(layout="indexed")
    assert t.interprete("bvh", "sm")
    t.build(pos, tri)
    lds_levels, spill_levels = t.profile_details()["lds_levels"], t.profile_details()["spill_levels"]
    assert t.depth > 40 and spill_levels > 0 and lds_levels + spill_levels == t.depth
    n = 1 << 20
    rays = scenes.stadium_rays(n, seed=3, sheets=40)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    # one launch over four batches, closest_hit and any_hit interleaved (each on the whole ray set, in halves)
    half = n // 2
    t.trace_batches_device([(d_rays[:half], d_hits[:half], "closest"), (d_rays[:half], d_occ[:half], "any"),
                            (d_rays[half:], d_occ[half:], "any"), (d_rays[half:], d_hits[half:], "closest")])
    torch.cuda.synchronize()
    hits, occ = _hits_of(d_hits), d_occ.cpu().numpy()
    t.profile_begin()
    d_hits2 = torch.empty_like(d_hits)
    t.closest_hit_device(d_rays, d_hits2, d_cnt)
    torch.cuda.synchronize()
    work = t.profile_work()
    pushes = t.profile_details()["spill_pushes"]
    t.profile_end()
    assert torch.equal(d_hits2, d_hits)
    o = orc.Bbvh(pos, tri, layout="indexed", split="sm")
    sample = np.arange(0, n, 16)
    want, want_counts, max_stack = o.closest_hit(rays[sample], counts=True, max_stack=True)
    assert_hits_equal(hits[sample], want, "deep-stack scene sample")
    assert (d_cnt.cpu().numpy().astype(np.uint32)[sample] == want_counts[:, 0:2]).all()
    assert (occ[sample] == o.any_hit(rays[sample])).all()
    beyond_lds = max_stack > lds_levels
    assert beyond_lds.mean() > 0.4 and max_stack.max() > 25     # most rays need the spill area, and more than 25 slots
    assert pushes >= 16 * int(beyond_lds.sum()) * 0.9            # every such ray pushed there at least once
    assert pushes < work[1]                                      # (a push follows an inner-node visit)
    valid = hits["t"] != FLT_MAX
    assert (occ == valid).all() and 0.3 < valid.mean() < 0.95
    # nothing lies in front of a closest hit -- up to the few ulp by which the reference's traversal itself may keep the
    # farther of two hits on coplanar neighbours (it enters a box only if its entry distance is BELOW the closest t so
    # far, bvh2.cpp:432-433), which a tessellated sheet is full of
    short = rays[valid].copy()
    short[:, 7] = hits["t"][valid] * np.float32(1.0 - 4e-6)
    assert (t.closest_hit(short)["t"] == FLT_MAX).all()
    t.close()
