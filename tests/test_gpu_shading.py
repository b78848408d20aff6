"""The stages after the tracer on the GPU -- shading of hits, framebuffer accumulation, 8-bit conversion --
against their numpy statements (raytracing_amd/wavefront.py), and a whole frame of the reference's
a3-brdf-test script (scripts/a3-brdf-test of "Assignment 3/rtgi-2021-a03-ref/") against the same frame computed
with the CPU oracle. Bit-exact except for the gamma curve (powf): at most one 8-bit step there."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from raytracing_amd import scenes, tracer, wavefront
from tests.util import assert_hits_equal

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "golden", "render-data", "brdf-test-noal.obj")
# camera and light of scripts/a3-brdf-test
CAM = dict(cam_pos=(-2.4008, 1.7537, -0.050779), cam_dir=(0.85235, -0.52249, -0.022323), cam_up=(0.52233, 0.85264, -0.013184))
LIGHT_POS, LIGHT_COL = (2.0, 1.5, 0.0), (4.0, 2.0, 1.0)
RMSE_8BIT = 0.25  # stated bound on the image difference, in 8-bit steps


def _scene():
    s = scenes.load_obj_materials(SCENE)
    kd, ks = s["kd"], s["ks"]
    luma = 0.212671 * kd[:, 0] + 0.715160 * kd[:, 1] + 0.072169 * kd[:, 2]
    albedo = np.where((luma > 1e-4)[:, None], kd, ks).astype(np.float32)
    s["albedo"] = np.power(albedo, np.float32(2.2), dtype=np.float32)  # libgi/scene.cpp:123-125
    return s


def _hits_of(d_hits):
    raw = d_hits.cpu().numpy()
    out = np.empty(len(raw), dtype=[("t", "f4"), ("beta", "f4"), ("gamma", "f4"), ("ref", "u4")])
    out["t"], out["beta"], out["gamma"] = raw[:, 0], raw[:, 1], raw[:, 2]
    out["ref"] = raw[:, 3].view(np.uint32)
    return out


@pytest.mark.parametrize("layout", ["flat", "indexed"])
def test_frame_of_the_brdf_test_script(layout, tmp_path):
    import torch
    s = _scene()
    w, h, spp = 320, 180, 3
    t = tracer.HipBvhTracer(layout=layout)
    t.build(s["pos"], s["tri"])
    with pytest.raises(tracer.RtgiHipError):  # shading before set_shading
        t.shade_primary_device(torch.zeros((4, 4), device="cuda"))
    t.set_shading(s["normals"], s["tri"], s["material_id"], s["albedo"])
    order = t.triangle_order() if layout == "flat" else np.arange(len(s["tri"]))
    tri, mat = s["tri"][order], s["material_id"][order]  # what hit ids index
    ns = wavefront.shading_normals(s["normals"], tri)
    o = orc.Bbvh(s["pos"], s["tri"], layout=layout)

    d_fb = torch.zeros((w * h, 4), dtype=torch.float32, device="cuda")
    d_fb_primary = torch.zeros((w * h, 4), dtype=torch.float32, device="cuda")
    fb = np.zeros((w * h, 4), np.float32)
    fb_primary = np.zeros((w * h, 4), np.float32)
    for sample in range(spp):
        jitter = np.stack([wavefront.uniform01(np.arange(w * h), 2 * sample, 9) - 0.5,
                           wavefront.uniform01(np.arange(w * h), 2 * sample + 1, 9) - 0.5], 1).astype(np.float32)
        # GPU: camera rays -> closest_hit -> shadow rays -> any_hit -> shading -> framebuffer
        d_jitter = t.pixel_offsets_device(w * h, sample, seed=9)
        assert np.array_equal(d_jitter.cpu().numpy(), jitter)
        d_rays = t.camera_rays_device(CAM["cam_dir"], CAM["cam_pos"], CAM["cam_up"], 65.0, w, h, d_offsets=d_jitter)
        d_hits = torch.empty((w * h, 4), dtype=torch.float32, device="cuda")
        t.closest_hit_device(d_rays, d_hits)
        d_shadow = t.shadow_rays_device(d_hits, LIGHT_POS)
        d_occ = torch.empty(w * h, dtype=torch.uint8, device="cuda")
        t.any_hit_device(d_shadow, d_occ)
        d_albedo = t.shade_primary_device(d_hits)
        d_rad = t.shade_local_device(d_hits, d_occ, LIGHT_POS, LIGHT_COL)
        t.framebuffer_add_device(d_fb_primary, d_albedo)
        t.framebuffer_add_device(d_fb, d_rad)
        # CPU: the same frame with the oracle as tracer
        rays = orc.cam_rays(CAM["cam_dir"], CAM["cam_pos"], CAM["cam_up"], 65.0, w, h, offsets=jitter)
        assert d_rays.cpu().numpy().tobytes() == rays.tobytes()
        hits = o.closest_hit(rays)
        assert_hits_equal(_hits_of(d_hits), hits, "primary hits, sample %d" % sample)
        valid = hits["t"] != wavefront.FLT_MAX
        shadow = np.zeros((w * h, 8), np.float32)
        shadow[:, 3], shadow[:, 6], shadow[:, 7] = 1.0, wavefront.EPS, -1.0  # empty interval for misses
        shadow[valid] = wavefront.shadow_rays(wavefront.hit_points(s["pos"], tri, hits[valid]), LIGHT_POS)
        occ = o.any_hit(shadow)
        assert (d_occ.cpu().numpy() == occ).all()
        albedo = wavefront.shade_primary(hits, mat, s["albedo"])
        rad = wavefront.shade_local(s["pos"], tri, ns, mat, s["albedo"], hits, occ, LIGHT_POS, LIGHT_COL)
        assert np.array_equal(d_albedo.cpu().numpy(), albedo)
        assert np.array_equal(d_rad.cpu().numpy().view(np.uint32), rad.view(np.uint32)), "radiance differs"
        assert (rad[:, 0:3].sum(1) > 0).mean() > 0.05  # the light reaches a fair part of the image
        fb_primary = wavefront.framebuffer_add(fb_primary, albedo)
        fb = wavefront.framebuffer_add(fb, rad)
    assert np.array_equal(d_fb.cpu().numpy().view(np.uint32), fb.view(np.uint32)), "framebuffer differs"
    assert np.array_equal(d_fb_primary.cpu().numpy().view(np.uint32), fb_primary.view(np.uint32))
    assert (fb[:, 3] == spp).all()
    for d_buf, buf in ((d_fb, fb), (d_fb_primary, fb_primary)):
        got = t.framebuffer_rgb8_device(d_buf, w, h).cpu().numpy()
        want = wavefront.framebuffer_rgb8(buf, w, h)
        diff = got.astype(np.int32) - want.astype(np.int32)
        assert np.abs(diff).max() <= 1
        assert np.sqrt((diff.astype(np.float64) ** 2).mean()) <= RMSE_8BIT
    from raytracing_amd import png
    path = str(tmp_path / "frame.png")
    png.write_rgb8(path, got)
    assert np.array_equal(png.read_rgb8(path), got)


def test_brdfs_against_their_host_statement():
    """Every BRDF of new_brdf (libgi/material.cpp:77-102) on the a3-brdf-test frame: lambert, gtr2 and layered-gtr2
    bit-identical to the numpy statement of libgi/material.cpp; the two with phong within 1e-5 relative (powf)."""
    import torch
    s = _scene()
    w, h = 400, 225
    t = tracer.HipBvhTracer(layout="indexed")
    t.build(s["pos"], s["tri"])
    t.set_shading(s["normals"], s["tri"], s["material_id"], s["albedo"])
    nm = len(s["albedo"])
    rs = np.random.RandomState(5)
    ns = wavefront.shading_normals(s["normals"], s["tri"])
    d_rays = t.camera_rays_device(CAM["cam_dir"], CAM["cam_pos"], CAM["cam_up"], 65.0, w, h)
    d_hits = torch.empty((w * h, 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_rays, d_hits)
    d_occ = torch.empty(w * h, dtype=torch.uint8, device="cuda")
    t.any_hit_device(t.shadow_rays_device(d_hits, LIGHT_POS), d_occ)
    rays, hits, occ = d_rays.cpu().numpy(), _hits_of(d_hits), d_occ.cpu().numpy()
    lambert = t.shade_local_device(d_hits, d_occ, LIGHT_POS, LIGHT_COL).cpu().numpy()
    assert np.array_equal(t.shade_local_brdf_device(d_rays, d_hits, d_occ, LIGHT_POS, LIGHT_COL).cpu().numpy().view(np.uint32),
                          lambert.view(np.uint32))  # set_shading leaves every material lambertian
    for kind, name in enumerate(wavefront.BRDF_NAMES):
        mats = dict(albedo=s["albedo"], ior=rs.uniform(1.2, 2.6, nm).astype(np.float32),
                    roughness=rs.uniform(0.02, 0.6, nm).astype(np.float32), kind=np.full(nm, kind))
        t.set_materials(mats["albedo"], mats["ior"], mats["roughness"], [name] * nm)
        got = t.shade_local_brdf_device(d_rays, d_hits, d_occ, LIGHT_POS, LIGHT_COL).cpu().numpy()
        want = wavefront.shade_local_brdf(s["pos"], s["tri"], ns, s["material_id"], mats, rays, hits, occ, LIGHT_POS, LIGHT_COL)
        assert np.isfinite(got).all() and (got[:, 0:3].sum(1) > 0).mean() > 0.05, name
        if "phong" in name:
            assert np.allclose(got, want, rtol=1e-5, atol=1e-30), name
        else:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
        if name != "lambert":
            assert not np.array_equal(got, lambert)
    with pytest.raises(tracer.RtgiHipError):
        t.set_materials(s["albedo"][:-1], np.ones(nm - 1), np.ones(nm - 1), ["lambert"] * (nm - 1))


def test_albedo_textures():
    """Materials with a texture take their albedo from it (diff_geom::albedo, libgi/rt.cpp:27-31; texture::sample,
    libgi/scene.h:24-32) in every shading stage, bit for bit like the numpy statement; dropping the textures brings
    the material colours back."""
    import torch
    s = _scene()
    w, h = 480, 270
    rs = np.random.RandomState(11)
    tex_coords = rs.uniform(-1.5, 2.5, (len(s["pos"]), 2)).astype(np.float32)  # beyond [0, 1): wraps
    textures = [rs.uniform(0, 1, (23, 37, 3)).astype(np.float32), rs.uniform(0, 1, (64, 64, 3)).astype(np.float32)]
    nm = len(s["albedo"])
    material_texture = np.array([(k % 3) - 1 for k in range(nm)], np.int32)  # -1, 0, 1, -1, ...
    for layout in ("flat", "indexed"):
        t = tracer.HipBvhTracer(layout=layout)
        t.build(s["pos"], s["tri"])
        t.set_shading(s["normals"], s["tri"], s["material_id"], s["albedo"])
        order = t.triangle_order() if layout == "flat" else np.arange(len(s["tri"]))
        tri, mat = s["tri"][order], s["material_id"][order]
        ns = wavefront.shading_normals(s["normals"], tri)
        d_rays = t.camera_rays_device(CAM["cam_dir"], CAM["cam_pos"], CAM["cam_up"], 65.0, w, h)
        d_hits = torch.empty((w * h, 4), dtype=torch.float32, device="cuda")
        t.closest_hit_device(d_rays, d_hits)
        d_occ = torch.zeros(w * h, dtype=torch.uint8, device="cuda")
        rays, hits, occ = d_rays.cpu().numpy(), _hits_of(d_hits), d_occ.cpu().numpy()
        plain = t.shade_primary_device(d_hits).cpu().numpy()
        t.set_textures(tex_coords, s["tri"], textures, material_texture)

        def albedo_fn(hv):
            return wavefront.albedo_of_hits(hv, mat, s["albedo"], tex_coords, tri, textures, material_texture)[1]
        valid, alb = wavefront.albedo_of_hits(hits, mat, s["albedo"], tex_coords, tri, textures, material_texture)
        want = np.zeros((w * h, 4), np.float32)
        want[valid, 0:3] = alb
        got = t.shade_primary_device(d_hits).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), layout
        assert not np.array_equal(got, plain)  # textured materials are in view
        want_l = wavefront.shade_local(s["pos"], tri, ns, mat, s["albedo"], hits, occ, LIGHT_POS, LIGHT_COL, albedo_fn=albedo_fn)
        got_l = t.shade_local_device(d_hits, d_occ, LIGHT_POS, LIGHT_COL).cpu().numpy()
        assert np.array_equal(got_l.view(np.uint32), want_l.view(np.uint32)), layout
        mats = dict(albedo=s["albedo"], ior=np.full(nm, 1.7, np.float32), roughness=np.full(nm, 0.2, np.float32), kind=np.full(nm, 3))
        t.set_materials(mats["albedo"], mats["ior"], mats["roughness"], ["gtr2"] * nm)
        want_b = wavefront.shade_local_brdf(s["pos"], tri, ns, mat, mats, rays, hits, occ, LIGHT_POS, LIGHT_COL, albedo_fn=albedo_fn)
        got_b = t.shade_local_brdf_device(d_rays, d_hits, d_occ, LIGHT_POS, LIGHT_COL).cpu().numpy()
        assert np.array_equal(got_b.view(np.uint32), want_b.view(np.uint32)), layout
        t.set_textures(tex_coords, s["tri"], [], material_texture)  # `material texture drop` for all
        assert np.array_equal(t.shade_primary_device(d_hits).cpu().numpy(), plain)
        with pytest.raises(tracer.RtgiHipError):
            t.set_textures(tex_coords, s["tri"], textures, np.full(nm, 2, np.int32))  # no such texture


def test_shading_argument_checks():
    import torch
    s = _scene()
    t = tracer.HipBvhTracer(layout="indexed")
    t.build(s["pos"], s["tri"])
    with pytest.raises(tracer.RtgiHipError):  # a different scene
        t.set_shading(s["normals"][:-3], s["tri"][:-1], s["material_id"][:-1], s["albedo"])
    bad = s["material_id"].copy()
    bad[0] = len(s["albedo"])
    with pytest.raises(tracer.RtgiHipError):  # material id out of range
        t.set_shading(s["normals"], s["tri"], bad, s["albedo"])
    t.set_shading(s["normals"], s["tri"], s["material_id"], s["albedo"])
    t.build(s["pos"], s["tri"])  # a rebuild drops the shading data
    with pytest.raises(tracer.RtgiHipError):
        t.shade_primary_device(torch.zeros((4, 4), device="cuda"))
