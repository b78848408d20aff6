"""Agreement of the three restated tracers (rt/seq, naive_bvh, binary_bvh_tracer in every layout and
builder) on identical rays -- the cross-check the reference's scripts do by swapping 'raytracer' lines
(scripts/a2-tris:6-8)."""
import numpy as np
import pytest

from oracle import oracle as orc
from raytracing_amd import scenes
from tests.util import FLT_MAX, assert_hits_equal, obj_scene, original_ids, rays_toward_scene

SMALL = ["tri.obj", "2tri.obj", "brdf-test-noal.obj", "mis.obj"]
VARIANTS = [
    dict(layout="flat", split="om"),
    dict(layout="indexed", split="om"),
    dict(layout="indexed", split="sm"),
    dict(layout="flat", split="sah", planes=8),
    dict(layout="indexed", split="sah", planes=16, max_tris=4),
    dict(layout="indexed", split="om", max_tris=3),
    dict(layout="indexed", esc=True, split="om"),
    dict(layout="indexed", esc=True, split="sah", planes=8),
]


def _check_scene(pos, tri, rays, variants=VARIANTS):
    seq = orc.Seq(pos, tri)
    want = seq.closest_hit(rays)
    want_any = seq.any_hit(rays)
    assert (want_any == (want["t"] != FLT_MAX)).all()

    nb = orc.NaiveBvh(pos, tri)
    got = nb.closest_hit(rays)
    got["ref"] = original_ids(got, nb.triangles())
    ties = _tie_mask(seq, rays, want)
    _assert_same_but_ties(got, want, ties, "naive_bvh")
    assert (nb.any_hit(rays) == want_any).all()

    for v in variants:
        b = orc.Bbvh(pos, tri, **v)
        got = b.closest_hit(rays)
        if v["layout"] == "flat":
            got["ref"] = original_ids(got, b.triangles())
        _assert_same_but_ties(got, want, ties, str(v))
        assert (b.any_hit(rays) == want_any).all(), v


def _tie_mask(seq, rays, want):
    """Rays whose closest t is reached by more than one triangle (exact ties): the winner then depends
    on encounter order, which differs between tracers."""
    valid = want["t"] != FLT_MAX
    ties = np.zeros(len(rays), bool)
    idx = np.flatnonzero(valid)
    if idx.size == 0:
        return ties
    # re-trace with t_max just above the found t and the found triangle excluded is not expressible
    # through the interface, so test by shrinking t_max to the found t: any_hit then must be false
    # unless another triangle has exactly... (strictly smaller t is impossible). Instead compare
    # against the reversed-order brute force.
    rev = orc.Seq(seq.pos, seq.tri[::-1].copy())
    other = rev.closest_hit(rays)
    other_ref = len(seq.tri) - 1 - other["ref"].astype(np.int64)
    ties[idx] = other_ref[idx] != want["ref"][idx]
    return ties


def _assert_same_but_ties(got, want, ties, what):
    """Bit-equal t, ref, beta, gamma -- except on (near-)ties. A BVH tracer prunes a child box whose entry
    distance is not below the closest t found so far (bvh2.cpp:432-433), so where two triangles are hit at
    distances one or two ulps apart (the shared diagonal of an axis-aligned quad) it may keep the first one
    while the brute-force loop keeps the smaller t. Those rays are excused, counted and bounded."""
    t_differs = got["t"].view(np.uint32) != want["t"].view(np.uint32)
    near = t_differs & (np.abs(got["t"].astype(np.float64) - want["t"]) <= 4e-7 * np.abs(want["t"].astype(np.float64)))
    assert (t_differs == near).all(), what
    assert near.sum() <= 0.01 * len(want), what
    ok = ~ties & ~near
    assert (got["ref"][ok] == want["ref"][ok]).all(), what
    valid = ok & (want["t"] != FLT_MAX)
    for k in ("beta", "gamma"):
        assert (got[k][valid].view(np.uint32) == want[k][valid].view(np.uint32)).all(), what


@pytest.mark.parametrize("name", SMALL)
def test_render_data_scenes(name):
    pos, tri, _ = obj_scene(name)
    rays = rays_toward_scene(pos, 4000, seed=11)
    variants = VARIANTS if len(tri) > 2 else [v for v in VARIANTS if not v.get("esc")]
    _check_scene(pos, tri, rays, variants)


def test_soup_2000():
    pos, tri, _ = scenes.soup(2000, seed=1)
    rays = scenes.random_rays(3000, seed=7)
    _check_scene(pos, tri, rays)


def test_cornell_and_terrain():
    pos, tri, _ = scenes.cornell(2)
    _check_scene(pos, tri, rays_toward_scene(pos, 3000, seed=3))
    pos, tri, _ = scenes.terrain_objects(3000, seed=3)
    _check_scene(pos, tri, rays_toward_scene(pos, 2000, seed=4))


def test_node_format_and_counts():
    pos, tri, _ = scenes.soup(1000, seed=2)
    b = orc.Bbvh(pos, tri, layout="indexed")
    nodes = b.nodes()
    assert len(nodes) == 2 * 1000 - 1 and b.root == 0
    leaf = nodes["link_r"] < 0
    assert leaf.sum() == 1000
    assert (nodes["link_r"][leaf] == -1).all()                       # one triangle per leaf
    assert sorted(-nodes["link_l"][leaf]) == list(range(1000))       # offsets cover the index once
    assert sorted(b.index()) == list(range(1000))
    rays = scenes.random_rays(500, seed=9)
    hits, counts, ms = b.closest_hit(rays, counts=True, max_stack=True)
    assert (counts[:, 0] >= 1).all() and (counts[:, 1] == counts[:, 2]).all()
    assert ms.max() <= b.depth + 1


def test_single_triangle_root_is_leaf():
    pos, tri, _ = obj_scene("tri.obj")
    b = orc.Bbvh(pos, tri)
    nodes = b.nodes()
    assert len(nodes) == 1 and nodes["link_l"][0] == 0 and nodes["link_r"][0] == -1
