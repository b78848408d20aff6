"""Parity of the HIP traversal with the CPU oracle, through the C ABI, on a real MI355X.

Bar: bit-exact. The kernels keep the reference's visiting order and arithmetic, so t, beta, gamma and
the triangle id must be identical to the oracle's for every ray, including exact ties; the per-ray
counts of inner nodes popped and triangles tested must be identical as well."""
import numpy as np
import pytest

from oracle import oracle as orc
from raytracing_amd import scenes, tracer
from tests.util import FLT_MAX, assert_hits_equal, obj_scene, rays_toward_scene

pytestmark = pytest.mark.gpu


def _configure(t, cmds):
    for c in cmds:
        assert t.interprete("bvh", c)


def _oracle_for(pos, tri, layout, esc, cmds):
    kw = dict(layout=layout, esc=esc, split="om", planes=0, max_tris=1)
    for c in cmds:
        p = c.split()
        if p[0] in ("om", "sm"):
            kw["split"] = p[0]
        elif p[0] == "sah":
            kw["split"], kw["planes"] = "sah", int(p[1])
        elif p[0] == "triangles":
            kw["max_tris"] = 1 if p[1] == "single" else int(p[2])
    return orc.Bbvh(pos, tri, **kw)


def _mixed_rays(pos, n, seed):
    """Rays aimed at the scene, rays from inside it, shadow-style rays with a finite t_max."""
    a = rays_toward_scene(pos, n // 2, seed)
    lo, hi = pos.min(0), pos.max(0)
    b = scenes.random_rays(n - n // 2, seed + 1)
    b[:, 0:3] = lo + (b[:, 0:3] + 10.0) / 20.0 * (hi - lo)
    rays = np.concatenate([a, b])
    rs = np.random.RandomState(seed + 2)
    short = rs.rand(len(rays)) < 0.3
    rays[short, 7] = rs.uniform(0.05, 1.0, size=short.sum()) * float(np.linalg.norm(hi - lo))
    return rays


def _check(pos, tri, rays, layout="flat", esc=False, cmds=()):
    t = tracer.HipBvhTracer(layout=layout, esc=esc)
    _configure(t, cmds)
    t.build(pos, tri)
    o = _oracle_for(pos, tri, layout, esc, cmds)
    assert t.nodes().tobytes() == o.nodes().tobytes()
    want, want_counts = o.closest_hit(rays, counts=True)
    got = t.closest_hit(rays)
    assert_hits_equal(got, want, "%s esc=%s %s" % (layout, esc, list(cmds)))
    want_any, want_any_counts = o.any_hit(rays, counts=True)
    assert (t.any_hit(rays) == want_any).all()
    _check_device_path(t, rays, want, want_counts, want_any, want_any_counts)
    t.close()


def _check_device_path(t, rays, want, want_counts, want_any, want_any_counts):
    import torch
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
    d_cnt = torch.zeros((len(rays), 2), dtype=torch.int32, device="cuda")
    t.closest_hit_device(d_rays, d_hits, d_cnt)
    torch.cuda.synchronize()
    got = d_hits.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)
    assert_hits_equal(got, want, "device path")
    cnt = d_cnt.cpu().numpy().astype(np.uint32)
    assert (cnt == want_counts[:, 0:2]).all()
    d_occ = torch.empty(len(rays), dtype=torch.uint8, device="cuda")
    d_cnt.zero_()
    t.any_hit_device(d_rays, d_occ, d_cnt)
    torch.cuda.synchronize()
    assert (d_occ.cpu().numpy() == want_any).all()
    assert (d_cnt.cpu().numpy().astype(np.uint32) == want_any_counts[:, 0:2]).all()


@pytest.mark.parametrize("name", ["tri.obj", "2tri.obj", "brdf-test-noal.obj", "mis.obj"])
def test_render_data_scenes(name):
    pos, tri, _ = obj_scene(name)
    rays = _mixed_rays(pos, 20000, seed=5)
    _check(pos, tri, rays, "flat")
    _check(pos, tri, rays, "indexed", cmds=["sah 4"])
    if len(tri) > 2:
        _check(pos, tri, rays, "indexed", esc=True)


VARIANTS = [
    ("flat", False, ()),
    ("indexed", False, ("sm",)),
    ("indexed", False, ("sah 16",)),
    ("indexed", False, ("sah 8", "triangles multiple 4")),
    ("flat", False, ("om", "triangles multiple 3")),
    ("indexed", True, ()),
]


@pytest.mark.parametrize("layout,esc,cmds", VARIANTS)
def test_generated_scenes(layout, esc, cmds):
    for pos, tri, _ in (scenes.soup(20000, seed=1), scenes.cornell(4), scenes.sponza_class(20000),
                        scenes.terrain_objects(20000)):
        _check(pos, tri, _mixed_rays(pos, 30000, seed=9), layout, esc, cmds)


def _chain_scene(n=60):
    """Triangles facing +x at x = -2^i with sizes growing alike: a spatial-median split peels off the
    leftmost triangle at every level, so the tree is a chain of depth n - 1."""
    tris = []
    for i in range(n):
        x, s = -(2.0 ** i), 0.25 * 2.0 ** i
        tris.append([[x, -s, -s], [x, s, -s], [x, 0.0, s]])
    return scenes._finish(np.array(tris))


def test_deep_tree_spills_the_lds_stack():
    # a chain-shaped tree: rays along -x enter every box, so the pending-node stack grows to the tree depth,
    # far beyond the 16 entries kept in LDS and beyond the reference's own 25 slots
    pos, tri, _ = _chain_scene(60)
    rs = np.random.RandomState(8)
    n = 5000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0] = 10.0
    rays[:, 1:3] = rs.uniform(-0.2, 0.2, size=(n, 2))
    d = np.concatenate([-np.ones((n, 1)), rs.uniform(-0.01, 0.01, size=(n, 2))], axis=1)
    d[::5] *= -1                                   # some rays point away and miss everything
    rays[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 6], rays[:, 7] = 1e-4, FLT_MAX
    rays[1::3, 0] = -rs.uniform(1, 2.0 ** 40, size=len(rays[1::3]))   # origins between the sheets
    t = tracer.HipBvhTracer(layout="indexed")
    t.interprete("bvh", "sm")
    t.build(pos, tri)
    assert t.depth >= 40
    o = orc.Bbvh(pos, tri, layout="indexed", split="sm")
    want, counts, max_stack = o.closest_hit(rays, counts=True, max_stack=True)
    assert max_stack.max() > 40
    assert (want["t"] != FLT_MAX).sum() > n // 2
    assert_hits_equal(t.closest_hit(rays), want, "chain tree")
    want_any, any_counts = o.any_hit(rays, counts=True)
    assert (t.any_hit(rays) == want_any).all()
    _check_device_path(t, rays, want, counts, want_any, any_counts)


def test_skewed_scene_with_unbalanced_trees():
    pos, tri, _ = scenes.skewed(60000, seed=5)
    rays = _mixed_rays(pos, 40000, seed=2)
    for cmds in (("sm",), ("sah 16",)):
        t = tracer.HipBvhTracer(layout="indexed")
        _configure(t, cmds)
        t.build(pos, tri)
        assert t.depth > 25
        o = _oracle_for(pos, tri, "indexed", False, cmds)
        assert_hits_equal(t.closest_hit(rays), o.closest_hit(rays), "skewed %s" % (cmds,))
        assert (t.any_hit(rays) == o.any_hit(rays)).all()


def test_axis_parallel_rays_and_origins_on_box_planes():
    # d components of exactly 0 give id = +-inf, and origins on box planes then give 0 * inf = NaN in the
    # slab test (libgi/intersect.h:210-228); the selects must resolve exactly like the reference's.
    pos, tri, _ = scenes.cornell(4)
    rs = np.random.RandomState(3)
    n = 20000
    rays = np.zeros((n, 8), np.float32)
    grid = np.linspace(-1, 1, 9)
    rays[:, 0:3] = rs.choice(grid, size=(n, 3))            # origins on wall and box planes
    axis = rs.randint(0, 3, n)
    kind = rs.randint(0, 3, n)
    d = rs.normal(size=(n, 3))
    for k in range(3):                                      # kind 0: one zero component, 1: two zeros
        d[(kind == 0) & (axis == k), k] = 0.0
        m = (kind == 1) & (axis == k)
        d[m, (k + 1) % 3] = 0.0
        d[m, (k + 2) % 3] = 0.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3:6] = d
    rays[:, 6], rays[:, 7] = 1e-4, FLT_MAX
    rays[::7, 3:6] = np.where(rays[::7, 3:6] == 0, -0.0, rays[::7, 3:6])   # negative zeros: id = -inf
    for layout, cmds in (("flat", ()), ("indexed", ("sah 8",))):
        _check(pos, tri, rays, layout, False, cmds)


def test_batch_sizes_and_argument_checks():
    import torch
    pos, tri, _ = scenes.soup(5000, seed=4)
    t = tracer.HipBvhTracer(layout="indexed")
    with pytest.raises(tracer.RtgiHipError):
        t.closest_hit(scenes.random_rays(8))               # nothing built yet
    t.build(pos, tri)
    o = orc.Bbvh(pos, tri, layout="indexed")
    assert len(t.closest_hit(np.zeros((0, 8), np.float32))) == 0
    for n in (1, 63, 64, 65, 255, 257, 4097, 100003):
        rays = scenes.random_rays(n, seed=n)
        assert_hits_equal(t.closest_hit(rays), o.closest_hit(rays), "n=%d" % n)
        assert (t.any_hit(rays) == o.any_hit(rays)).all()
    # a ray buffer that is not 16-byte aligned is refused, not read
    buf = torch.zeros(8 * 16 + 1, dtype=torch.float32, device="cuda")
    rc = tracer.lib().rtgi_hip_closest_hit_device(t._h, buf.data_ptr() + 4, 16, buf.data_ptr(), None, None)
    assert rc == -1


def test_large_batches_are_reordered_without_changing_results():
    # batches of 2^16 rays and more are sorted (direction octant, length class, Morton code of the origin) before
    # tracing when their origins are scattered ("auto"), always ("on") or never ("off"), and each XCD starts on
    # its own part; hits, any-hit bytes and per-ray counts must not notice
    import torch
    for scene, cmds in ((scenes.soup(20000, seed=8, size=0.2), ()), (scenes.skewed(30000, seed=6), ("sm",))):
        pos, tri, _ = scene
        t = tracer.HipBvhTracer(layout="indexed")
        _configure(t, cmds)
        t.build(pos, tri)
        o = _oracle_for(pos, tri, "indexed", False, cmds)
        rays = _mixed_rays(pos, 300000, seed=21)
        want, want_counts = o.closest_hit(rays, counts=True)
        want_any, want_any_counts = o.any_hit(rays, counts=True)
        assert_hits_equal(t.closest_hit(rays), want, "closest_hit, reorder auto")
        assert t.interprete("hip", "reorder off")
        assert_hits_equal(t.closest_hit(rays), want, "closest_hit in given order")
        # coherent rays (one origin, neighbouring directions): "auto" keeps their order, "on" sorts them anyway
        coherent = rays.copy()
        coherent[:, 0:3] = pos.mean(0)
        coherent = coherent[np.lexsort((coherent[:, 5], coherent[:, 4], coherent[:, 3]))]
        want_coherent = o.closest_hit(coherent)
        for mode in ("auto", "on"):
            assert t.interprete("hip", "reorder " + mode)
            assert_hits_equal(t.closest_hit(coherent), want_coherent, "coherent rays, reorder " + mode)
            assert (t.any_hit(rays) == want_any).all()
            _check_device_path(t, rays, want, want_counts, want_any, want_any_counts)
        with pytest.raises(tracer.RtgiHipError):
            t.interprete("hip", "reorder sometimes")
        # the figure "auto" decides by: mean inner-node pops of probe rays, same order of magnitude as these rays'
        assert 0.2 * want_counts[:, 0].mean() < t.incoherent_ray_cost < 5.0 * want_counts[:, 0].mean()


def test_upload_of_a_foreign_tree():
    pos, tri, _ = scenes.soup(8000, seed=6)
    o = orc.Bbvh(pos, tri, layout="indexed", split="sah", planes=8)
    t = tracer.HipBvhTracer(layout="indexed")
    t.upload_bvh(o.nodes(), o.root, o.index(), pos, tri)
    rays = scenes.random_rays(20000, seed=12)
    assert_hits_equal(t.closest_hit(rays), o.closest_hit(rays), "uploaded tree")


def test_waves_of_one_direction_octant(monkeypatch):
    # Waves whose rays all look the same way along every axis take the near and far planes of a box by position
    # instead of by min / max (sorted batches and neighbouring camera rays make such waves). That is only exact for
    # boxes with min <= max: a tree from outside with a box turned inside out must fall back to the general test,
    # which treats it like the box it mirrors (intersect.h:212-217 selects by comparing the products).
    pos, tri, _ = scenes.soup(30000, seed=16, size=0.3)
    o = orc.Bbvh(pos, tri, layout="indexed")
    rng = np.random.default_rng(3)
    rays = scenes.random_rays(200000, seed=33)
    for octant in range(8):  # 8192 rays per octant on purpose: whole waves of each
        part = rays[octant * 8192:(octant + 1) * 8192]
        for k in range(3):
            part[:, 3 + k] = np.abs(part[:, 3 + k]) * (-1.0 if (octant >> k) & 1 else 1.0)
    want, want_counts = o.closest_hit(rays, counts=True)
    want_any, want_any_counts = o.any_hit(rays, counts=True)
    t = tracer.HipBvhTracer(layout="indexed")
    t.build(pos, tri)
    for mode in ("off", "on"):
        assert t.interprete("hip", "reorder " + mode)
        assert_hits_equal(t.closest_hit(rays), want, "octant waves, reorder " + mode)
        _check_device_path(t, rays, want, want_counts, want_any, want_any_counts)
    # the same tree with some boxes turned inside out along one axis
    nodes = o.nodes().copy()
    inner = np.flatnonzero(nodes["link_r"] >= 0)
    for i in rng.choice(inner, 200, replace=False):
        box = nodes["box_l" if i & 1 else "box_r"][i]
        k = int(i) % 3
        box[k], box[3 + k] = box[3 + k], box[k]
    results = []
    for general_only in (True, False):
        if general_only:
            monkeypatch.setenv("RTGI_HIP_NO_OCTANT_WAVES", "1")
        else:
            monkeypatch.delenv("RTGI_HIP_NO_OCTANT_WAVES")
        u = tracer.HipBvhTracer(layout="indexed")
        u.upload_bvh(nodes, o.root, o.index(), pos, tri)
        assert u.interprete("hip", "reorder on")
        results.append((u.closest_hit(rays), u.any_hit(rays)))
    assert_hits_equal(results[1][0], results[0][0], "inside-out boxes")
    assert (results[1][1] == results[0][1]).all()
    # (min / max do not see which plane was called min: the mirrored boxes give the original tree's hits)
    assert_hits_equal(results[0][0], want, "inside-out boxes against the oracle's tree")


def test_full_size_soup_properties():
    # benchmark-size scene (2^20 triangles): oracle on a sample, size-independent properties on the rest
    import torch
    pos, tri, _ = scenes.soup(1 << 20, seed=1)
    t = tracer.HipBvhTracer(layout="indexed")
    t.build(pos, tri)
    n = 1 << 20
    rays = scenes.random_rays(n, seed=7)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    t.closest_hit_device(d_rays, d_hits)
    t.any_hit_device(d_rays, d_occ)
    torch.cuda.synchronize()
    hits = d_hits.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)
    occ = d_occ.cpu().numpy()
    valid = hits["t"] != FLT_MAX
    assert (occ == valid).all()                                     # any_hit <=> closest_hit finds something
    assert 0.3 < valid.mean() < 0.7
    assert (hits["beta"][valid] > 0).all() and (hits["gamma"][valid] > 0).all()
    assert (hits["beta"][valid] + hits["gamma"][valid] <= 1).all()
    assert (hits["ref"][valid] < len(tri)).all()
    # shortening every hit ray to just before its hit must turn it into a miss (nothing closer exists)
    short = rays[valid].copy()
    short[:, 7] = hits["t"][valid]
    assert (t.closest_hit(short)["t"] == FLT_MAX).all()
    # the barycentric point lies on the ray: A + beta (B-A) + gamma (C-A) == o + t d
    v = pos[tri[hits["ref"][valid]]].astype(np.float64)
    b, g = hits["beta"][valid, None].astype(np.float64), hits["gamma"][valid, None].astype(np.float64)
    p_tri = v[:, 0] * (1 - b - g) + v[:, 1] * b + v[:, 2] * g
    p_ray = rays[valid, 0:3] + hits["t"][valid, None].astype(np.float64) * rays[valid, 3:6]
    assert np.abs(p_tri - p_ray).max() < 1e-3
    # oracle on a sample of the same rays
    o = orc.Bbvh(pos, tri, layout="indexed")
    sample = np.arange(0, n, 64)
    assert_hits_equal(hits[sample], o.closest_hit(rays[sample]), "2^20 soup sample")


@pytest.mark.parametrize("layout,esc,cmds", [("flat", False, ()), ("indexed", False, ("sah 8", "triangles multiple 2")),
                                             ("indexed", True, ("sm",))])
def test_rays_aimed_at_shared_vertices_and_edges(layout, esc, cmds):
    # meshes with shared vertices and edges: a ray aimed exactly at a vertex or an edge midpoint grazes up to six
    # triangles, whose tests differ in the last bit or tie exactly; which one wins depends on the visiting order
    rs = np.random.RandomState(17)
    for pos, tri, _ in (scenes.cornell(3), scenes.terrain_objects(6000), scenes.sponza_class(9000)):
        lo, hi = pos.min(0), pos.max(0)
        n = 12000
        corners = pos[tri[rs.randint(0, len(tri), n)]]                      # (n, 3, 3)
        k = rs.randint(0, 3, n)
        vertex = corners[np.arange(n), k]
        edge_mid = ((corners[np.arange(n), k] + corners[np.arange(n), (k + 1) % 3]) * np.float32(0.5)).astype(np.float32)
        target = np.where((rs.rand(n) < 0.5)[:, None], vertex, edge_mid)
        origin = (lo + rs.rand(n, 3) * (hi - lo)).astype(np.float32)
        origin[: n // 3] = vertex[rs.permutation(n)[: n // 3]]                 # a third start exactly on a vertex
        d = (target - origin).astype(np.float32)
        norm = np.linalg.norm(d, axis=1, keepdims=True)
        keep = norm[:, 0] > 1e-6
        rays = np.empty((int(keep.sum()), 8), np.float32)
        rays[:, 0:3] = origin[keep]
        rays[:, 3:6] = (d[keep] / norm[keep]).astype(np.float32)
        rays[:, 6] = 1e-4
        rays[:, 7] = FLT_MAX
        _check(pos, tri, rays, layout, esc, cmds)


def test_multi_million_triangle_deep_tree_mixed_batches():
    # 2^22 triangles of very uneven size and density, spatial-median split: a tree far deeper than the part of
    # the traversal stack that lives in LDS (and than the reference's 25 slots). closest_hit and any_hit batches
    # alternate on one handle; oracle (growable stack) on a sample, including per-ray pop / test counts;
    # properties on everything.
    import torch
    pos, tri, _ = scenes.skewed(1 << 22, seed=5)
    t = tracer.HipBvhTracer(layout="indexed")
    _configure(t, ["sm"])
    t.build(pos, tri)
    assert t.depth > 40
    n = 1 << 20
    rays = _mixed_rays(pos, n, seed=31)
    d_rays = torch.from_numpy(rays).cuda()
    halves = (slice(0, n // 2), slice(n // 2, n))
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    for rep in range(2):  # closest, any, closest, any ... on alternating halves, then the other way round
        for k, part in enumerate(halves):
            if (k + rep) % 2 == 0:
                t.closest_hit_device(d_rays[part], d_hits[part], d_cnt[part])
            else:
                t.any_hit_device(d_rays[part], d_occ[part])
    torch.cuda.synchronize()
    hits = d_hits.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)
    occ = d_occ.cpu().numpy()
    valid = hits["t"] != FLT_MAX
    assert (occ == valid).all()
    assert 0.05 < valid.mean() < 0.95
    short = rays[valid].copy()
    short[:, 7] = hits["t"][valid]
    assert (t.closest_hit(short)["t"] == FLT_MAX).all()
    o = orc.Bbvh(pos, tri, layout="indexed", split="sm")
    sample = np.arange(0, n, 16)
    want, want_counts, max_stack = o.closest_hit(rays[sample], counts=True, max_stack=True)
    assert_hits_equal(hits[sample], want, "deep tree sample")
    assert (d_cnt.cpu().numpy().astype(np.uint32)[sample] == want_counts[:, 0:2]).all()
    assert (occ[sample] == o.any_hit(rays[sample])).all()
    # (depth 63 selects the kernel variant with the spill area; these rays pile up far fewer pending nodes than the
    # tree is deep -- the chain-shaped scene above is the one that fills the spill area)
    assert 8 <= max_stack.max() <= t.depth


@pytest.mark.parametrize("scene,cmds,n", [("soup", (), 150000), ("skewed", ("sm",), 40000)])
def test_several_batches_in_one_launch(scene, cmds, n):
    # closest_hit and any_hit batches of different sizes (one of them empty) share one launch; every ray's result
    # must be what the single calls give, i.e. the oracle's. The large case goes through the ray sort, the small
    # one (below 2^16 rays in total) does not; the skewed scene uses the kernel variant with the spill area.
    import torch
    pos, tri, _ = scenes.soup(30000, seed=3, size=0.2) if scene == "soup" else scenes.skewed(40000, seed=6)
    t = tracer.HipBvhTracer(layout="indexed")
    _configure(t, cmds)
    t.build(pos, tri)
    assert t.interprete("hip", "reorder on")
    o = _oracle_for(pos, tri, "indexed", False, cmds)
    rays = _mixed_rays(pos, n, seed=41)
    cuts = [0, n // 3, n // 3, n // 2 + 11, n]  # the second batch is empty
    kinds = ["closest", "any", "any", "closest"]
    d_all = torch.from_numpy(rays).cuda()
    batches, outs = [], []
    for k, kind in enumerate(kinds):
        part = d_all[cuts[k]:cuts[k + 1]].contiguous()
        m = part.shape[0]
        d_out = torch.empty((m, 4), dtype=torch.float32, device="cuda") if kind == "closest" else \
            torch.full((m,), 7, dtype=torch.uint8, device="cuda")
        batches.append((part, d_out, kind))
        outs.append(d_out)
    t.trace_batches_device(batches)
    torch.cuda.synchronize()
    for k, kind in enumerate(kinds):
        part = rays[cuts[k]:cuts[k + 1]]
        if kind == "closest":
            got = outs[k].cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)
            assert_hits_equal(got, o.closest_hit(part), "batch %d" % k)
        else:
            assert (outs[k].cpu().numpy() == o.any_hit(part)).all(), "batch %d" % k
    t.trace_batches_device([])  # nothing to do
    with pytest.raises(tracer.RtgiHipError):
        t.trace_batches_device(batches * 9)  # more than 32
    # thirty-two batches, the most a launch takes (the frame level traces the shadow and bounce rays of sixteen samples together)
    edges = [n * k // 32 for k in range(33)]
    many, many_out = [], []
    for k in range(32):
        part = d_all[edges[k]:edges[k + 1]].contiguous()
        m = edges[k + 1] - edges[k]
        d_out = torch.empty((m, 4), dtype=torch.float32, device="cuda") if k % 2 == 0 else torch.full((m,), 7, dtype=torch.uint8, device="cuda")
        many.append((part, d_out, "closest" if k % 2 == 0 else "any"))
        many_out.append(d_out)
    t.trace_batches_device(many)
    torch.cuda.synchronize()
    want_hits, want_any = o.closest_hit(rays), o.any_hit(rays)
    for k in range(32):
        if k % 2 == 0:
            assert_hits_equal(many_out[k].cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1), want_hits[edges[k]:edges[k + 1]], "batch %d of 32" % k)
        else:
            assert (many_out[k].cpu().numpy() == want_any[edges[k]:edges[k + 1]]).all(), "batch %d of 32" % k


def test_launches_on_different_streams_do_not_overlap_on_the_scratch():
    # one handle, calls issued back to back on alternating streams without any synchronisation in between: the
    # library orders them on the GPU (they share ray counters and sort buffers), so every result is right
    import torch
    pos, tri, _ = scenes.soup(30000, seed=3, size=0.2)
    t = tracer.HipBvhTracer(layout="flat")
    t.build(pos, tri)
    assert t.interprete("hip", "reorder on")
    o = _oracle_for(pos, tri, "flat", False, ())
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    sets = [_mixed_rays(pos, 70000 + 5000 * k, seed=50 + k) for k in range(6)]
    d_sets = [torch.from_numpy(r).cuda() for r in sets]
    d_hits = [torch.empty((len(r), 4), dtype=torch.float32, device="cuda") for r in sets]
    d_occ = [torch.empty(len(r), dtype=torch.uint8, device="cuda") for r in sets]
    torch.cuda.synchronize()
    for k in range(len(sets)):
        t.closest_hit_device(d_sets[k], d_hits[k], stream=streams[k % 3])
        t.any_hit_device(d_sets[k], d_occ[k], stream=streams[(k + 1) % 3])
    torch.cuda.synchronize()
    for k, rays in enumerate(sets):
        got = d_hits[k].cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)
        assert_hits_equal(got, o.closest_hit(rays), "set %d" % k)
        assert (d_occ[k].cpu().numpy() == o.any_hit(rays)).all()


def test_stored_hit_vectors():
    # the committed vectors of tests/golden/hit_vectors.npz, without the oracle in the loop
    import os
    import torch
    from tests.util import GOLDEN, HIT_VECTOR_CASES, hit_vector_rays
    stored = np.load(os.path.join(GOLDEN, "hit_vectors.npz"))
    for k, (name, layout, split, planes) in enumerate(HIT_VECTOR_CASES):
        pos, tri, _ = obj_scene(name)
        rays = hit_vector_rays(pos, k)
        t = tracer.HipBvhTracer(layout=layout)
        if split == "sah":
            assert t.interprete("bvh", "sah %d" % planes)
        t.build(pos, tri)
        d_rays = torch.from_numpy(rays).cuda()
        d_hits = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
        d_occ = torch.empty(len(rays), dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros((len(rays), 2), dtype=torch.int32, device="cuda")
        t.closest_hit_device(d_rays, d_hits, d_cnt)
        key = "%s/%s/%s" % (name, layout, split)
        got = d_hits.cpu().numpy().view(np.uint32)
        want = stored[key + "/hits"]
        hit = want[:, 0] != np.float32(FLT_MAX).view(np.uint32)
        assert np.array_equal(got[:, [0, 3]], want[:, [0, 3]]), key          # t bits and triangle id, every ray
        assert np.array_equal(got[hit], want[hit]), key                       # barycentrics where there is a hit
        assert np.array_equal(d_cnt.cpu().numpy().astype(np.uint32), stored[key + "/counts"]), key
        d_cnt.zero_()
        t.any_hit_device(d_rays, d_occ, d_cnt)
        assert np.array_equal(d_occ.cpu().numpy(), stored[key + "/occluded"]), key
        assert np.array_equal(d_cnt.cpu().numpy().astype(np.uint32), stored[key + "/any_counts"]), key
        t.close()


def test_degenerate_triangles_and_pathological_rays():
    # zero-area and collinear triangles (M = 0 in Cramer's rule: inf / NaN, which every comparison rejects,
    # libgi/intersect.h:43-60), duplicated triangles (exact ties), and rays the renderer would never build on purpose:
    # zero direction, NaN / inf components, empty and inverted intervals, t_max = +inf. Nothing may hang or fault and
    # every answer must be the oracle's.
    rs = np.random.RandomState(23)
    pos, tri, _ = scenes.soup(6000, seed=12, size=0.4)
    pos = pos.copy()
    v = pos.reshape(-1, 3, 3)
    v[0:200, 1] = v[0:200, 0]                                  # two corners coincide
    v[200:400, 2] = (v[200:400, 0] + v[200:400, 1]) * 0.5      # collinear
    v[400:600] = v[400:600, 0:1]                               # a point
    v[600:900] = v[900:1200]                                   # exact duplicates of other triangles
    pos = v.reshape(-1, 3)
    n = 30000
    rays = _mixed_rays(pos, n, seed=77)
    k = rs.randint(0, 12, n)
    rays[k == 0, 3:6] = 0.0                                    # no direction at all: id = inf everywhere
    rays[k == 1, 3] = np.nan
    rays[k == 2, 0] = np.inf
    rays[k == 3, 7] = np.inf                                   # unbounded ray (not FLT_MAX)
    rays[k == 4, 6], rays[k == 4, 7] = 5.0, 1.0                # inverted interval
    rays[k == 5, 6] = np.nan
    rays[k == 6, 7] = np.nan
    rays[k == 7, 3:6] *= np.float32(1e-30)                     # unnormalised, tiny: id overflows towards inf
    rays[k == 8, 6] = -1.0                                     # may hit behind the origin
    for layout, esc, cmds in (("flat", False, ()), ("indexed", False, ("sah 8", "triangles multiple 3")), ("indexed", False, ("sm",))):
        _check(pos, tri, rays, layout, esc, cmds)
    # the same rays, three times over, through the sorted path (keys of NaN / inf rays must be harmless)
    t = tracer.HipBvhTracer(layout="indexed")
    t.build(pos, tri)
    assert t.interprete("hip", "reorder on")
    big = np.concatenate([rays, rays[::-1], rays])
    o = orc.Bbvh(pos, tri, layout="indexed")
    want, want_counts = o.closest_hit(big, counts=True)
    want_any, want_any_counts = o.any_hit(big, counts=True)
    _check_device_path(t, big, want, want_counts, want_any, want_any_counts)


def test_handle_on_another_gpu_than_the_current_one():
    """Build, probe and trace on GPU 1 while GPU 0 is the caller's current device (what `raytracer hip device 1` does
    in a host program that never selects a device): every buffer of the handle must live on its own GPU."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    torch.cuda.set_device(0)
    pos, tri, _ = scenes.soup(6000, seed=4)  # more than 4096 leaf slots: the probe rays of "hip reorder auto" run
    t = tracer.HipBvhTracer(layout="indexed", device=1)
    t.build(pos, tri)
    assert torch.cuda.current_device() == 0 and t.incoherent_ray_cost > 0
    rays = rays_toward_scene(pos, 40000, 3)
    o = orc.Bbvh(pos, tri, layout="indexed")
    assert_hits_equal(t.closest_hit(rays), o.closest_hit(rays), "host batch on GPU 1")
    assert np.array_equal(t.any_hit(rays), o.any_hit(rays))
    assert torch.cuda.current_device() == 0


def _page_locked_copy(a):
    """(pointer, numpy view) of a page-locked copy of a C-contiguous array."""
    import ctypes
    L = tracer.lib()
    ptr = L.rtgi_hip_host_alloc(max(a.nbytes, 16))
    assert ptr
    view = np.frombuffer((ctypes.c_char * a.nbytes).from_address(ptr), dtype=a.dtype).reshape(a.shape)
    view[...] = a
    return ptr, view


@pytest.mark.parametrize("scene,cmds,reorder,chunk_order", [("soup", (), "on", "bins"), ("soup", (), "on", "sort"), ("soup", (), "off", "bins"),
                                                            ("skewed", ("sm",), "auto", "bins"), ("skewed", ("sm",), "auto", "sort")])
def test_page_locked_host_batches_in_one_launch(scene, cmds, reorder, chunk_order, monkeypatch):
    """Host batches in page-locked memory go through the GPU in one launch that takes up the later chunks of the batch
    while it runs (each chunk ordered on its own, by octant and length class in two kernels or by the full key with the
    radix sort): results are those of the device entry points, ray by ray."""
    import torch
    monkeypatch.setenv("RTGI_HIP_STREAM_COARSE", "1" if chunk_order == "bins" else "0")
    pos, tri, _ = scenes.soup(60000, seed=2, size=0.2) if scene == "soup" else scenes.skewed(60000, seed=6)
    t = tracer.HipBvhTracer(layout="indexed")
    _configure(t, cmds)
    t.build(pos, tri)
    assert t.interprete("hip", "reorder " + reorder)
    n = 700001
    rays = _mixed_rays(pos, n, seed=33)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    t.closest_hit_device(d_rays, d_hits)
    t.any_hit_device(d_rays, d_occ)
    torch.cuda.synchronize()
    want, want_any = d_hits.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1), d_occ.cpu().numpy()
    L = tracer.lib()
    p_rays, v_rays = _page_locked_copy(rays)
    p_hits, v_hits = _page_locked_copy(np.zeros(n, tracer.HIT_DTYPE))
    p_occ, v_occ = _page_locked_copy(np.full(n, 7, np.uint8))
    try:
        for _ in range(2):  # (twice: the second call finds the streams, events and buffers of the first)
            v_hits.view(np.uint8)[...] = 0xAB
            t._check(L.rtgi_hip_closest_hit(t._h, p_rays, n, p_hits))
            assert_hits_equal(v_hits, want, "page-locked closest_hit")
            t._check(L.rtgi_hip_any_hit(t._h, p_rays, n, p_occ))
            assert (v_occ == want_any).all()
        assert t.host_path_stats() == dict(one_launch=4, gave_up=0)
        o = _oracle_for(pos, tri, "indexed", False, cmds)
        sample = np.arange(0, n, 8)
        assert_hits_equal(v_hits[sample], o.closest_hit(rays[sample]), "page-locked closest_hit against the oracle")
    finally:
        for ptr in (p_rays, p_hits, p_occ):
            L.rtgi_hip_host_free(ptr)
    t.close()


def test_page_locked_host_batch_whose_launch_gives_up(monkeypatch):
    """A one-launch host batch whose launch does not see the later parts of the batch arrive in time (here: it waits for
    10 nanoseconds) ends without them, says so, and the batch is traced chunk by chunk: same results, no hang; a handle
    to which that happens twice goes back to chunked launches for good."""
    import torch
    monkeypatch.setenv("RTGI_HIP_STREAM_WAIT_TICKS", "1")
    monkeypatch.setenv("RTGI_HIP_STREAM_PARTS", "0.02,0.5,1")  # (a first part that is used up long before the second is there)
    pos, tri, _ = scenes.soup(60000, seed=2, size=0.2)
    t = tracer.HipBvhTracer(layout="indexed")
    t.build(pos, tri)
    assert t.interprete("hip", "reorder on")
    n = 900000
    rays = _mixed_rays(pos, n, seed=34)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_rays, d_hits)
    torch.cuda.synchronize()
    want = d_hits.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)
    L = tracer.lib()
    p_rays, _ = _page_locked_copy(rays)
    p_hits, v_hits = _page_locked_copy(np.zeros(n, tracer.HIT_DTYPE))
    try:
        for call in range(3):  # after the second such batch the handle stops trying
            v_hits.view(np.uint8)[...] = 0xAB
            t._check(L.rtgi_hip_closest_hit(t._h, p_rays, n, p_hits))
            assert_hits_equal(v_hits, want, "page-locked closest_hit after the launch gave up")
            assert t.host_path_stats() == dict(one_launch=min(call + 1, 2), gave_up=min(call + 1, 2))
    finally:
        L.rtgi_hip_host_free(p_rays), L.rtgi_hip_host_free(p_hits)
    t.close()


def test_page_locked_host_batch_when_streams_cannot_run_side_by_side(monkeypatch):
    """Before its first one-launch batch a handle tries whether a kernel on its second stream runs while one on the launch
    stream waits for it; where it does not (streams sharing a hardware queue; here: forced), the handle never starts a
    launch that would wait in vain, and page-locked batches are traced in chunks as before."""
    import torch
    monkeypatch.setenv("RTGI_HIP_STREAM_PROBE_FAIL", "1")
    pos, tri, _ = scenes.soup(60000, seed=2, size=0.2)
    t = tracer.HipBvhTracer(layout="indexed")
    t.build(pos, tri)
    n = 600000
    rays = _mixed_rays(pos, n, seed=35)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_rays, d_hits)
    torch.cuda.synchronize()
    want = d_hits.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1)
    L = tracer.lib()
    p_rays, _ = _page_locked_copy(rays)
    p_hits, v_hits = _page_locked_copy(np.zeros(n, tracer.HIT_DTYPE))
    try:
        for _ in range(2):
            v_hits.view(np.uint8)[...] = 0xAB
            t._check(L.rtgi_hip_closest_hit(t._h, p_rays, n, p_hits))
            assert_hits_equal(v_hits, want, "page-locked closest_hit, chunked")
        assert t.host_path_stats() == dict(one_launch=0, gave_up=0)
    finally:
        L.rtgi_hip_host_free(p_rays), L.rtgi_hip_host_free(p_hits)
    t.close()


@pytest.mark.parametrize("n", [3000, 700001])
def test_page_locked_buffers_that_are_not_16_byte_aligned(n):
    """Page-locked memory a caller registered may start anywhere: rays and hit records 4 bytes off the 16-byte grid are
    traced like any others (the results then come down by a copy instead of being stored by the kernel)."""
    import ctypes
    pos, tri, _ = scenes.soup(20000, seed=8, size=0.3)
    t = tracer.HipBvhTracer(layout="flat")
    t.build(pos, tri)
    rays = _mixed_rays(pos, n, seed=36)
    want = t.closest_hit(rays)
    want_any = t.any_hit(rays)
    L = tracer.lib()
    p_rays, p_hits, p_occ = L.rtgi_hip_host_alloc(rays.nbytes + 16), L.rtgi_hip_host_alloc(n * 16 + 16), L.rtgi_hip_host_alloc(n + 16)
    assert p_rays and p_hits and p_occ
    try:
        ctypes.memmove(p_rays + 4, rays.ctypes.data, rays.nbytes)
        ctypes.memset(p_hits, 0xAB, n * 16 + 16)
        t._check(L.rtgi_hip_closest_hit(t._h, p_rays + 4, n, p_hits + 4))
        got = np.frombuffer(ctypes.string_at(p_hits + 4, n * 16), dtype=tracer.HIT_DTYPE)
        assert_hits_equal(got, want, "unaligned page-locked closest_hit")
        assert ctypes.string_at(p_hits, 4) == b"\xab" * 4 and ctypes.string_at(p_hits + 4 + n * 16, 12) == b"\xab" * 12
        t._check(L.rtgi_hip_any_hit(t._h, p_rays + 4, n, p_occ + 3))
        assert np.array_equal(np.frombuffer(ctypes.string_at(p_occ + 3, n), dtype=np.uint8), want_any)
    finally:
        for ptr in (p_rays, p_hits, p_occ):
            L.rtgi_hip_host_free(ptr)
    t.close()
