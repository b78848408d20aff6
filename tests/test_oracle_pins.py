"""The CPU oracle against everything the reference tree itself holds for this path:
the two known-answer cases in its driver sources and the camera-ray dump of Assignment 2."""
import gzip
import os

import numpy as np
import pytest

from oracle import oracle as orc

FLT_MAX = np.finfo(np.float32).max


def _ray(o, d, t_min=1e-4, t_max=FLT_MAX):
    return np.array(list(o) + list(d) + [t_min, t_max], np.float32)


def test_triangle_known_answer():
    # Assignment 1/driver/main.cpp:88-109: ray (0,0,3)->(3,2,-3) against (1,1,0),(5,1,0),(3,4,0) hits.
    hit, t, beta, gamma = orc.tri_intersect([1, 1, 0], [5, 1, 0], [3, 4, 0], _ray([0, 0, 3], [3, 2, -3]))
    assert hit
    assert t == pytest.approx(1.0, rel=1e-6)
    assert beta == pytest.approx(1 / 3, rel=1e-6)
    assert gamma == pytest.approx(1 / 3, rel=1e-6)


def test_box_known_answer():
    # Assignment 2/rtgi-2021-a02/driver/main.cpp:69-97: same ray against the flat box y=10 misses.
    r = _ray([0, 0, 3], [3, 2, -3])
    for variant in (0, 4):
        hit, _ = orc.box_intersect([-5, 10, -5], [5, 10, 5], r, variant)
        assert not hit
    # a box the ray does pass through is reported with its entry distance by both variants
    h0, e0 = orc.box_intersect([2, 1, -1], [4, 3, 1], r, 0)
    h4, e4 = orc.box_intersect([2, 1, -1], [4, 3, 1], r, 4)
    assert h0 and h4
    assert e0 == pytest.approx(2 / 3, rel=1e-6) and e4 == pytest.approx(2 / 3, rel=1e-6)


def test_triangle_edges_are_misses():
    # libgi/intersect.h:59-60: beta > 0 and gamma > 0 are strict, beta + gamma <= 1 is not.
    a, b, c = [0, 0, 0], [1, 0, 0], [0, 1, 0]
    assert not orc.tri_intersect(a, b, c, _ray([0.5, 0, 1], [0, 0, -1]))[0]   # on edge ab: gamma == 0
    assert not orc.tri_intersect(a, b, c, _ray([0, 0.5, 1], [0, 0, -1]))[0]   # on edge ac: beta == 0
    assert orc.tri_intersect(a, b, c, _ray([0.5, 0.5, 1], [0, 0, -1]))[0]     # on edge bc: beta+gamma == 1
    assert not orc.tri_intersect(a, b, c, _ray([0.25, 0.25, 1], [0, 0, -1], t_max=1.0))[0]  # t < t_max strict
    assert not orc.tri_intersect(a, b, c, _ray([0.25, 0.25, 1], [0, 0, 1]))[0]              # behind the origin


def test_cam_rays_match_assignment2_dump(golden_dir):
    # Assignment 2/rtgi-2021-a02/test.obj was written by test_camrays (libgi/camera.cpp:24-34) for the
    # a2-sibenik camera: at -13 -12 0, look 1 0 0, up 0 1 0, fovy 65, 1280x720, every 10th pixel.
    with gzip.open(os.path.join(golden_dir, "a2_camrays_test.obj.gz"), "rt") as f:
        lines = f.read().split("\n")
    vs = [l for l in lines if l.startswith("v ")]
    assert len(vs) == 2 * 9216
    rays = orc.cam_rays([1, 0, 0], [-13, -12, 0], [0, 1, 0], 65.0, 1280, 720, stride=10)
    assert len(rays) == 9216
    mismatched = 0
    for i in range(9216):
        want_o, want_d = vs[2 * i], vs[2 * i + 1]
        got_o = "v %g %g %g" % tuple(float(x) for x in rays[i, 0:3])
        got_d = "v %g %g %g" % tuple(float(x) for x in rays[i, 3:6])
        assert got_o == want_o
        if got_d != want_d:
            # the dump keeps 6 significant digits; allow a last-digit difference, but count it
            w = np.array([float(x) for x in want_d.split()[1:]])
            assert np.allclose(w, rays[i, 3:6], rtol=0, atol=1.5e-6)
            mismatched += 1
    assert mismatched == 0


def test_oracle_reproduces_the_stored_hit_vectors():
    # tests/golden/hit_vectors.npz (written by tests/golden/make_hit_vectors.py): the oracle's own answers, frozen
    from tests.util import GOLDEN, HIT_VECTOR_CASES, hit_vector_rays, obj_scene
    stored = np.load(os.path.join(GOLDEN, "hit_vectors.npz"))
    for k, (name, layout, split, planes) in enumerate(HIT_VECTOR_CASES):
        pos, tri, _ = obj_scene(name)
        rays = hit_vector_rays(pos, k)
        o = orc.Bbvh(pos, tri, layout=layout, split=split, planes=planes)
        hits, counts = o.closest_hit(rays, counts=True)
        occluded, any_counts = o.any_hit(rays, counts=True)
        key = "%s/%s/%s" % (name, layout, split)
        assert np.array_equal(hits.view(np.uint32).reshape(-1, 4), stored[key + "/hits"]), key
        assert np.array_equal(counts[:, 0:2], stored[key + "/counts"]), key
        assert np.array_equal(occluded, stored[key + "/occluded"]), key
        assert np.array_equal(any_counts[:, 0:2], stored[key + "/any_counts"]), key
