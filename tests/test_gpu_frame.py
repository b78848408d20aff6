"""The frame level (include/rtgi_hip_frame.h, host side in C++): `raytracer hip devices N` + `commit` + `run` of the
reference's a3-brdf-test script ("Assignment 3/rtgi-2021-a03-ref/scripts/a3-brdf-test") against the same frame
computed with the CPU oracle as tracer and the numpy statements of the stages; and the several-GPU path rehearsed
with several handles on the one GPU of the box (tiles: bit-identical to one handle; samples: sums in another order)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from raytracing_amd import frame, scenes, wavefront

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "golden", "render-data", "brdf-test-noal.obj")
CAM = dict(cam_pos=(-2.4008, 1.7537, -0.050779), cam_dir=(0.85235, -0.52249, -0.022323), cam_up=(0.52233, 0.85264, -0.013184))
LIGHT_POS, LIGHT_COL = (2.0, 1.5, 0.0), (4.0, 2.0, 1.0)


def _scene():
    s = scenes.load_obj_materials(SCENE)
    kd, ks = s["kd"], s["ks"]
    luma = 0.212671 * kd[:, 0] + 0.715160 * kd[:, 1] + 0.072169 * kd[:, 2]
    albedo = np.where((luma > 1e-4)[:, None], kd, ks).astype(np.float32)
    s["albedo"] = np.power(albedo, np.float32(2.2), dtype=np.float32)  # libgi/scene.cpp:123-125
    return s


def _renderer(s, devices=(0,), layout="flat"):
    r = frame.HipRenderer(devices=devices, layout=layout)
    r.build(s["pos"], s["normals"], s["tri"], s["material_id"], s["albedo"])
    return r


def _oracle_frame(s, layout, order, w, h, spp, seed, algo):
    tri, mat = s["tri"][order], s["material_id"][order]
    ns = wavefront.shading_normals(s["normals"], tri)
    o = orc.Bbvh(s["pos"], s["tri"], layout=layout)
    fb = np.zeros((w * h, 4), np.float32)
    for sample in range(spp):
        jitter = np.stack([wavefront.uniform01(np.arange(w * h), 2 * sample, seed) - 0.5,
                           wavefront.uniform01(np.arange(w * h), 2 * sample + 1, seed) - 0.5], 1).astype(np.float32)
        rays = orc.cam_rays(CAM["cam_dir"], CAM["cam_pos"], CAM["cam_up"], 65.0, w, h, offsets=jitter)
        hits = o.closest_hit(rays)
        if algo == "primary":
            rad = wavefront.shade_primary(hits, mat, s["albedo"])
        else:
            valid = hits["t"] != wavefront.FLT_MAX
            shadow = np.zeros((w * h, 8), np.float32)
            shadow[:, 3], shadow[:, 6], shadow[:, 7] = 1.0, wavefront.EPS, -1.0
            shadow[valid] = wavefront.shadow_rays(wavefront.hit_points(s["pos"], tri, hits[valid]), LIGHT_POS)
            occ = o.any_hit(shadow)
            rad = wavefront.shade_local(s["pos"], tri, ns, mat, s["albedo"], hits, occ, LIGHT_POS, LIGHT_COL)
        fb = wavefront.framebuffer_add(fb, rad)
    return fb


@pytest.mark.parametrize("layout,algo", [("flat", "local"), ("indexed", "local"), ("flat", "primary")])
def test_frame_against_the_oracle_traced_frame(layout, algo):
    s = _scene()
    w, h, spp, seed = 320, 180, 3, 9
    r = _renderer(s, layout=layout)
    fb, rgb, stats = r.render(fovy=65.0, w=w, h=h, sppx=spp, algo=algo, light_pos=LIGHT_POS, light_col=LIGHT_COL, seed=seed, **CAM)
    assert stats["n_devices"] == 1 and stats["rays"] == w * h * spp * (2 if algo == "local" else 1)
    from raytracing_amd import tracer
    import ctypes as C
    n_index = tracer.lib().rtgi_hip_num_index(C.c_void_p(frame.lib().rtgi_hip_renderer_handle(r._r, 0)))
    order = np.zeros(n_index, np.uint32)
    tracer.lib().rtgi_hip_get_triangle_order(C.c_void_p(frame.lib().rtgi_hip_renderer_handle(r._r, 0)), order.ctypes.data)
    want = _oracle_frame(s, layout, order if layout == "flat" else np.arange(len(s["tri"])), w, h, spp, seed, algo)
    assert np.array_equal(fb.view(np.uint32), want.view(np.uint32)), "framebuffer differs from the oracle-traced frame"
    assert (fb[:, 3] == spp).all()
    want8 = wavefront.framebuffer_rgb8(want, w, h)
    diff = rgb.astype(np.int32) - want8.astype(np.int32)
    assert np.abs(diff).max() <= 1 and np.sqrt((diff.astype(np.float64) ** 2).mean()) <= 0.25  # powf: one 8-bit step


CORNELL_LIGHT = (0.0, 0.8, 0.3)


def _cornell():
    pos, tri, nrm = scenes.cornell(3)
    rs = np.random.RandomState(0)
    # shading normals on the side of the light (local_illumination lights only that side, gi/primary-hit.cpp:46-47)
    to_light = np.asarray(CORNELL_LIGHT, np.float32) - pos
    nrm = np.where((np.einsum("ij,ij->i", nrm, to_light) < 0)[:, None], -nrm, nrm).astype(np.float32)
    return dict(pos=pos, tri=tri, normals=nrm, material_id=rs.randint(0, 3, len(tri)).astype(np.uint32),
                albedo=np.array([[0.7, 0.7, 0.7], [0.6, 0.1, 0.1], [0.1, 0.6, 0.1]], np.float32))


CORNELL_CAM = dict(cam_pos=(0.0, 0.0, 3.2), cam_dir=(0.0, 0.0, -1.0), cam_up=(0.0, 1.0, 0.0))


@pytest.mark.parametrize("algo,bounces", [("local", 0), ("path", 3)])
def test_tiles_over_several_handles_are_bit_identical(algo, bounces):
    w, h, spp = 250, 130, 2  # not a multiple of the tile edge
    if algo == "local":
        s = _scene()
        kw = dict(fovy=65.0, w=w, h=h, sppx=spp, algo=algo, bounces=bounces, light_pos=LIGHT_POS, light_col=LIGHT_COL, seed=4, **CAM)
    else:  # a closed room: paths go on
        s = _cornell()
        kw = dict(fovy=40.0, w=w, h=h, sppx=spp, algo=algo, bounces=bounces, light_pos=CORNELL_LIGHT, light_col=(1.0, 1.0, 1.0), seed=4,
                  **CORNELL_CAM)
    one, one8, st1 = _renderer(s).render(**kw)
    three = _renderer(s, devices=(0, 0, 0))
    fb, rgb, st3 = three.render(sharding="tiles", tile=32, **kw)
    assert st3["n_devices"] == 3 and st3["used_rccl"] == 0  # one GPU named three times: added up by a kernel
    assert np.array_equal(fb.view(np.uint32), one.view(np.uint32))
    assert np.array_equal(rgb, one8)
    assert st3["rays"] == st1["rays"]
    if algo == "path":
        assert st1["rays"] > 5 * w * h * spp  # paths went on after the first hit
        local, _, _ = _renderer(s).render(**dict(kw, algo="local"))
        assert (one[:, 0:3] >= local[:, 0:3]).all()  # bounces add light: the first vertex is that of `algo local`
        assert one[:, 0:3].sum(dtype=np.float64) > 1.2 * local[:, 0:3].sum(dtype=np.float64)
    fb2, _, _ = three.render(sharding="tiles", tile=0, **kw)  # second frame on the same renderer, default tile edge
    assert np.array_equal(fb2.view(np.uint32), one.view(np.uint32))


@pytest.mark.parametrize("algo,bounces,spp", [("primary", 0, 5), ("local", 0, 11), ("path", 3, 11)])
def test_samples_traced_in_groups_give_the_frame_of_samples_traced_one_by_one(algo, bounces, spp, monkeypatch):
    # a part with few pixels traces several samples per launch (a launch over fewer rays than the GPU holds is all tail):
    # groups of up to 31 samples, here of at most 1, 3, 8 (balanced: 6 + 5) and all 11; the framebuffer must not notice
    w, h = 190, 100
    s = _cornell()
    kw = dict(fovy=40.0, w=w, h=h, sppx=spp, algo=algo, bounces=bounces, light_pos=CORNELL_LIGHT, light_col=(1.0, 1.0, 1.0), seed=9, **CORNELL_CAM)
    frames = {}
    for most in ("1", "3", "8", "31"):
        monkeypatch.setenv("RTGI_HIP_FRAME_GROUP", most)
        frames[most] = _renderer(s).render(**kw)
    for most in ("3", "8", "31"):
        assert np.array_equal(frames[most][0].view(np.uint32), frames["1"][0].view(np.uint32)), most
        assert np.array_equal(frames[most][1], frames["1"][1])
        assert frames[most][2]["rays"] == frames["1"][2]["rays"]
    assert (frames["1"][0][:, 3] == spp).all()
    monkeypatch.setenv("RTGI_HIP_FRAME_GROUP", "31")
    two = _renderer(s, devices=(0, 0)).render(sharding="samples", **kw)  # each part groups its own samples
    assert np.allclose(two[0][:, 0:3], frames["1"][0][:, 0:3], rtol=2e-6, atol=1e-7)


def test_deep_tree_mixed_batches_over_two_handles():
    """A tree deeper than the LDS part of the stack (skewed scene, spatial-median splits) under `algo path`, whose
    launches mix any_hit and closest_hit batches: two handles (tiles) render the frame of one, bit for bit."""
    pos, tri, nrm = scenes.skewed(200000, seed=5)
    s = dict(pos=pos, tri=tri, normals=nrm, material_id=np.zeros(len(tri), np.uint32), albedo=np.array([[0.6, 0.6, 0.6]], np.float32))
    cam = dict(cam_pos=(0.0, 0.0, 2.0), cam_dir=(0.0, 0.0, -1.0), cam_up=(0.0, 1.0, 0.0))  # the dense middle of the scene fills the view
    kw = dict(fovy=40.0, w=192, h=128, sppx=2, algo="path", bounces=2, light_pos=(0.5, 2.0, 1.5), light_col=(5.0, 5.0, 5.0), seed=8, **cam)
    frames = []
    for devices in ((0,), (0, 0)):
        r = frame.HipRenderer(devices=devices, layout="indexed")
        assert r.interprete("bvh", "sm")
        r.build(s["pos"], s["normals"], s["tri"], s["material_id"], s["albedo"])
        from raytracing_amd import tracer
        import ctypes as C
        depth = tracer.lib().rtgi_hip_depth(C.c_void_p(frame.lib().rtgi_hip_renderer_handle(r._r, 0)))
        assert depth > 24, "the scene is meant to need the spill area (depth %d)" % depth
        fb, _, st = r.render(sharding="tiles", **kw)
        frames.append((fb, st))
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32))
    assert frames[0][1]["rays"] == frames[1][1]["rays"] > 2.5 * 192 * 128 * 2  # most paths go on after the camera ray
    assert (frames[0][0][:, 0:3].sum(1) > 0).mean() > 0.02


def test_samples_over_several_handles():
    s = _scene()
    w, h, spp = 200, 120, 5
    kw = dict(fovy=65.0, w=w, h=h, sppx=spp, algo="local", light_pos=LIGHT_POS, light_col=LIGHT_COL, seed=2, **CAM)
    one, _, _ = _renderer(s).render(**kw)
    fb, _, st = _renderer(s, devices=(0, 0)).render(sharding="samples", **kw)
    assert (fb[:, 3] == spp).all()
    assert np.allclose(fb[:, 0:3], one[:, 0:3], rtol=2e-6, atol=1e-7)  # means formed from sums in another order
    assert st["rays"] == 2 * w * h * spp


def test_one_process_per_gpu_form_goes_through_rccl():
    """rank 0 of a world of 1: the frame is that of the plain renderer, and the framebuffer did pass through
    ncclAllReduce (the call every rank makes when there are several)."""
    s = _scene()
    kw = dict(fovy=65.0, w=160, h=90, sppx=2, algo="local", light_pos=LIGHT_POS, light_col=LIGHT_COL, seed=3, **CAM)
    one, _, _ = _renderer(s).render(**kw)
    r = frame.HipRenderer(devices=(0,), rank=0, world=1, unique_id=frame.HipRenderer.unique_id())
    r.build(s["pos"], s["normals"], s["tri"], s["material_id"], s["albedo"])
    for sharding in ("tiles", "samples"):
        fb, _, st = r.render(sharding=sharding, **kw)
        assert st["used_rccl"] == 1 and st["n_devices"] == 1
        assert np.array_equal(fb.view(np.uint32), one.view(np.uint32))


def test_frame_argument_checks():
    from raytracing_amd import tracer
    s = _scene()
    with pytest.raises(tracer.RtgiHipError):
        frame.HipRenderer(devices=())
    with pytest.raises(tracer.RtgiHipError):
        frame.HipRenderer(devices=(0,), layout="flat", esc=True)  # refused like the registry does
    r = frame.HipRenderer()
    with pytest.raises(tracer.RtgiHipError):  # run before commit
        r.render(fovy=65.0, w=8, h=8, **CAM)
    r.build(s["pos"], s["normals"], s["tri"], s["material_id"], s["albedo"])
    with pytest.raises(tracer.RtgiHipError):
        r.render(fovy=65.0, w=0, h=8, **CAM)
    with pytest.raises(tracer.RtgiHipError):
        r.render(fovy=65.0, w=8, h=8, sppx=0, **CAM)
    assert r.interprete("bvh", "sah 8") and not r.interprete("whatever")


def test_a_failing_part_is_an_error_not_a_hang(monkeypatch):
    """One part of several fails while the others render: render() returns the error (no part enters the framebuffer
    exchange), and the renderer goes on working afterwards."""
    from raytracing_amd import tracer
    s = _scene()
    kw = dict(fovy=65.0, w=96, h=64, sppx=1, algo="local", light_pos=LIGHT_POS, light_col=LIGHT_COL, seed=1, **CAM)
    r = _renderer(s, devices=(0, 0))
    good, _, _ = r.render(**kw)
    monkeypatch.setenv("RTGI_HIP_FAIL_PART", "1")
    with pytest.raises(tracer.RtgiHipError, match="part 1.*injected"):
        r.render(**kw)
    monkeypatch.delenv("RTGI_HIP_FAIL_PART")
    again, _, _ = r.render(**kw)
    assert np.array_equal(again.view(np.uint32), good.view(np.uint32))


def test_a_failing_rank_takes_part_in_the_status_exchange(monkeypatch):
    """One process per GPU (here a world of one rank, through RCCL): a rank whose part fails still enters the one-word
    status all-reduce, returns its own error, and the communicator survives for the next frame."""
    from raytracing_amd import tracer
    s = _scene()
    kw = dict(fovy=65.0, w=96, h=64, sppx=1, algo="local", light_pos=LIGHT_POS, light_col=LIGHT_COL, seed=1, **CAM)
    r = frame.HipRenderer(devices=(0,), rank=0, world=1, unique_id=frame.HipRenderer.unique_id())
    r.build(s["pos"], s["normals"], s["tri"], s["material_id"], s["albedo"])
    good, _, st = r.render(**kw)
    assert st["used_rccl"] == 1
    monkeypatch.setenv("RTGI_HIP_FAIL_PART", "0")
    with pytest.raises(tracer.RtgiHipError, match="injected"):
        r.render(**kw)
    monkeypatch.delenv("RTGI_HIP_FAIL_PART")
    again, _, st = r.render(**kw)
    assert st["used_rccl"] == 1 and np.array_equal(again.view(np.uint32), good.view(np.uint32))


def test_shared_build_gives_every_handle_the_tree_of_a_single_build():
    """rtgi_hip_build_shared (what `commit` does for `devices N`): one host build, every handle ends up with the nodes,
    the triangle order and the hits of a handle built on its own."""
    from raytracing_amd import tracer
    pos, tri, _ = scenes.soup(30000, seed=3)
    rays = scenes.random_rays(20000, seed=4)
    alone = tracer.HipBvhTracer(layout="indexed")
    alone.interprete("bvh", "sah 8")
    alone.build(pos, tri)
    want = alone.closest_hit(rays)
    group = [tracer.HipBvhTracer(layout="indexed") for _ in range(3)]
    for t in group:
        t.interprete("bvh", "sah 8")
    tracer.HipBvhTracer.build_shared(group, pos, tri)
    for t in group:
        assert t.nodes().tobytes() == alone.nodes().tobytes()
        assert np.array_equal(t.triangle_order(), alone.triangle_order())
        assert t.incoherent_ray_cost == alone.incoherent_ray_cost
        assert t.closest_hit(rays).tobytes() == want.tobytes()
    group[1].interprete("bvh", "om")
    with pytest.raises(tracer.RtgiHipError, match="same layout and bvh settings"):
        tracer.HipBvhTracer.build_shared(group, pos, tri)


def test_rt_bench_leaves_the_callers_device_alone():
    import torch
    s = _scene()
    r = _renderer(s)
    n, ms = r.rt_bench(fovy=65.0, w=320, h=200, **CAM)
    assert n == 320 * 200 and ms > 0.0
    assert torch.cuda.current_device() == 0
