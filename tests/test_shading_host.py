"""Host-side pieces around the shading stages: OBJ/MTL reading, the numpy statements of framebuffer::add and
framebuffer::png (libgi/framebuffer.cpp:10-30 of "Assignment 3/rtgi-2021-a03-ref/"), the PNG writer."""
import os

import numpy as np

from raytracing_amd import png, scenes, wavefront

HERE = os.path.dirname(os.path.abspath(__file__))


def test_obj_with_materials():
    s = scenes.load_obj_materials(os.path.join(HERE, "golden", "render-data", "brdf-test-noal.obj"))
    assert s["tri"].shape == (16, 3) and s["pos"].shape == (48, 3) and s["normals"].shape == (48, 3)
    assert len(s["materials"]) == 8 and s["material_id"].max() == 7
    assert np.allclose(np.linalg.norm(s["normals"], axis=1), 1.0, atol=1e-3)
    assert np.allclose(s["kd"][s["materials"].index("diffuse")], [0.100117, 0.405654, 0.103297])
    same_positions, _, _ = scenes.load_obj(os.path.join(HERE, "golden", "render-data", "brdf-test-noal.obj"))
    assert np.array_equal(same_positions, s["pos"])


def test_framebuffer_is_a_running_mean():
    rs = np.random.RandomState(3)
    samples = rs.uniform(0, 2, size=(5, 1000, 4)).astype(np.float32)
    fb = np.zeros((1000, 4), np.float32)
    for s in samples:
        fb = wavefront.framebuffer_add(fb, s)
    assert (fb[:, 3] == 5).all()
    assert np.allclose(fb[:, 0:3], samples[:, :, 0:3].mean(0), rtol=1e-5)


def test_rgb8_conversion_and_png_round_trip(tmp_path):
    w, h = 7, 5
    fb = np.zeros((w * h, 4), np.float32)
    fb[:, 0] = np.linspace(-0.5, 1.5, w * h)  # clamps at both ends
    fb[:, 1] = 0.5
    fb[:, 2] = 1.0
    rgb = wavefront.framebuffer_rgb8(fb, w, h)
    assert rgb.shape == (h, w, 3) and rgb.dtype == np.uint8
    assert rgb[-1, 0, 0] == 0 and rgb[0, -1, 0] == 255  # rows flipped: framebuffer row 0 is the bottom image row
    assert (rgb[..., 1] == int(0.5 ** (1 / 2.2) * 255)).all() and (rgb[..., 2] == 255).all()
    path = str(tmp_path / "x.png")
    png.write_rgb8(path, rgb)
    assert open(path, "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    assert np.array_equal(png.read_rgb8(path), rgb)
