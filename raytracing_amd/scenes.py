"""Scene inputs for the tracer: a small OBJ reader and seeded synthetic scene generators.

The reference loads scenes through assimp (libgi/scene.cpp:88-175), which is not available here, and its
large scenes (sibenik, sponza) are not part of the tree; the generators below provide scenes of the same
class and size.  Every generator returns ``(pos, tri, nrm)``: float32 vertex positions (nv, 3), uint32
vertex ids per triangle (nt, 3) and float32 vertex normals (nv, 3).  Like assimp without vertex joining,
every triangle owns three vertices of its own.
"""
import os

import numpy as np


def _finish(tri_pos, nrm=None):
    """tri_pos: (nt, 3, 3) float -> (pos, tri, nrm) with three private vertices per triangle."""
    tri_pos = np.ascontiguousarray(tri_pos, dtype=np.float32)
    nt = tri_pos.shape[0]
    pos = tri_pos.reshape(nt * 3, 3)
    tri = np.arange(nt * 3, dtype=np.uint32).reshape(nt, 3)
    if nrm is None:
        e1 = tri_pos[:, 1] - tri_pos[:, 0]
        e2 = tri_pos[:, 2] - tri_pos[:, 0]
        n = np.cross(e1.astype(np.float64), e2.astype(np.float64))
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        n = np.where(ln > 0, n / np.maximum(ln, 1e-300), np.array([0.0, 1.0, 0.0]))
        nrm = np.repeat(n.astype(np.float32), 3, axis=0)
    return pos, tri, np.ascontiguousarray(nrm, dtype=np.float32)


def load_obj(path):
    """Minimal Wavefront OBJ reader: 'v' and 'f' records, polygons fan-triangulated."""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "f":
                ids = [int(tok.split("/")[0]) for tok in p[1:]]
                ids = [i - 1 if i > 0 else len(verts) + i for i in ids]
                for k in range(1, len(ids) - 1):
                    faces.append([ids[0], ids[k], ids[k + 1]])
    v = np.asarray(verts, dtype=np.float32)
    fa = np.asarray(faces, dtype=np.int64)
    return _finish(v[fa])


def load_obj_materials(path):
    """OBJ reader that keeps what shading needs: positions, normals ('vn', zero if a corner has none), the material
    of every triangle ('usemtl') and, from the 'mtllib' beside the file if present, Kd / Ks per material.
    Returns dict(pos (3 nt, 3), tri (nt, 3), normals (3 nt, 3), material_id (nt,), materials [names], kd, ks)."""
    verts, vnorms, faces, fnorms, fmat, names = [], [], [], [], [], []
    mtl_file = None
    current = 0
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append([float(x) for x in p[1:4]])
            elif p[0] == "vn":
                vnorms.append([float(x) for x in p[1:4]])
            elif p[0] == "mtllib":
                mtl_file = os.path.join(os.path.dirname(path), p[1])
            elif p[0] == "usemtl":
                if p[1] not in names:
                    names.append(p[1])
                current = names.index(p[1])
            elif p[0] == "f":
                corners = []
                for tok in p[1:]:
                    parts = tok.split("/")
                    vi = int(parts[0])
                    ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                    corners.append((vi - 1 if vi > 0 else len(verts) + vi, ni - 1 if ni > 0 else (len(vnorms) + ni if ni else -1)))
                for k in range(1, len(corners) - 1):
                    faces.append([corners[0][0], corners[k][0], corners[k + 1][0]])
                    fnorms.append([corners[0][1], corners[k][1], corners[k + 1][1]])
                    fmat.append(current)
    if not names:
        names = ["default"]
    v = np.asarray(verts, np.float32)
    vn = np.vstack([np.asarray(vnorms, np.float32).reshape(-1, 3), np.zeros((1, 3), np.float32)])  # index -1 -> zero
    pos = v[np.asarray(faces, np.int64)].reshape(-1, 3)
    normals = vn[np.asarray(fnorms, np.int64)].reshape(-1, 3)
    tri = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
    kd, ks = np.zeros((len(names), 3), np.float32), np.zeros((len(names), 3), np.float32)
    if mtl_file and os.path.exists(mtl_file):
        cur = None
        with open(mtl_file) as f:
            for line in f:
                p = line.split()
                if not p:
                    continue
                if p[0] == "newmtl":
                    cur = names.index(p[1]) if p[1] in names else None
                elif cur is not None and p[0] in ("Kd", "kd"):
                    kd[cur] = [float(x) for x in p[1:4]]
                elif cur is not None and p[0] in ("Ks", "ks"):
                    ks[cur] = [float(x) for x in p[1:4]]
    return dict(pos=np.ascontiguousarray(pos), tri=tri, normals=np.ascontiguousarray(normals),
                material_id=np.asarray(fmat, np.uint32), materials=names, kd=kd, ks=ks)


def soup(n_tris=1 << 20, seed=1, extent=10.0, size=0.05):
    """Random small triangles: centres U[-extent, extent]^3, vertex offsets U[-size, size]^3."""
    rs = np.random.RandomState(seed)
    c = rs.uniform(-extent, extent, size=(n_tris, 1, 3))
    off = rs.uniform(-size, size, size=(n_tris, 3, 3))
    return _finish((c + off).astype(np.float32))


def _quad_grid(origin, du, dv, nu, nv, displace=None):
    """Tessellate the parallelogram origin + s*du + t*dv (s, t in [0,1]) into 2*nu*nv triangles."""
    origin, du, dv = (np.asarray(a, dtype=np.float64) for a in (origin, du, dv))
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, 1.0, nv + 1)
    S, T = np.meshgrid(s, t, indexing="ij")
    P = origin + S[..., None] * du + T[..., None] * dv
    if displace is not None:
        P = P + displace(S, T)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.stack([a, b, c], axis=2).reshape(-1, 3, 3)
    t2 = np.stack([a, c, d], axis=2).reshape(-1, 3, 3)
    return np.concatenate([t1, t2], axis=0)


def _box(lo, hi, n=1):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    e = hi - lo
    ex, ey, ez = np.diag(e)
    return np.concatenate([
        _quad_grid(lo, ex, ey, n, n), _quad_grid(lo + ez, ey, ex, n, n),
        _quad_grid(lo, ey, ez, n, n), _quad_grid(lo + ex, ez, ey, n, n),
        _quad_grid(lo, ez, ex, n, n), _quad_grid(lo + ey, ex, ez, n, n)], axis=0)


def _cylinder(base, radius, height, n_around, n_up):
    base = np.asarray(base, float)

    def disp(S, T):
        ang = 2 * np.pi * S
        return np.stack([radius * np.cos(ang), 0 * S, radius * np.sin(ang)], axis=-1)
    return _quad_grid(base, [0, 0, 0], [0, height, 0], n_around, n_up, displace=disp)


def _sphere(centre, radius, n):
    centre = np.asarray(centre, float)

    def disp(S, T):
        th, ph = np.pi * T, 2 * np.pi * S
        return radius * np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], axis=-1)
    return _quad_grid(centre, [0, 0, 0], [0, 0, 0], 2 * n, n, displace=disp)


def cornell(n=1):
    """Five walls and two boxes; 10*n*n + 24*n*n triangles (34 for n=1). Camera looks down -z from z>0."""
    parts = [
        _quad_grid([-1, -1, -1], [2, 0, 0], [0, 0, 2], n, n),   # floor
        _quad_grid([-1, 1, -1], [0, 0, 2], [2, 0, 0], n, n),    # ceiling
        _quad_grid([-1, -1, -1], [0, 2, 0], [2, 0, 0], n, n),   # back
        _quad_grid([-1, -1, -1], [0, 0, 2], [0, 2, 0], n, n),   # left
        _quad_grid([1, -1, -1], [0, 2, 0], [0, 0, 2], n, n),    # right
        _box([-0.7, -1, -0.6], [-0.1, 0.2, 0.0], n),
        _box([0.15, -1, -0.1], [0.7, -0.4, 0.5], n),
    ]
    return _finish(np.concatenate(parts, axis=0))


def sponza_class(target_tris=262144, seed=0):
    """Atrium-like architectural scene (floor, walls, two colonnades with arches, hanging drapes) in
    [-1500,1500] x [0,1200] x [-700,700], tessellated to target_tris triangles within a few percent (the
    tessellation density is searched for the closest count)."""
    def build(k, count_only):
        g = lambda a: max(1, int(round(a * k)))  # noqa: E731
        rs = np.random.RandomState(seed)
        parts, count = [], [0]

        def grid(origin, du, dv, nu, nv, displace=None):
            count[0] += 2 * nu * nv
            if not count_only:
                parts.append(_quad_grid(origin, du, dv, nu, nv, displace=displace))

        grid([-1500, 0, -700], [3000, 0, 0], [0, 0, 1400], g(24), g(12),
             displace=lambda S, T: np.stack([0 * S, 4 * np.sin(40 * S) * np.sin(31 * T), 0 * S], -1))
        grid([-1500, 1200, -700], [0, 0, 1400], [3000, 0, 0], g(6), g(12))
        grid([-1500, 0, -700], [0, 1200, 0], [3000, 0, 0], g(8), g(20))
        grid([-1500, 0, 700], [3000, 0, 0], [0, 1200, 0], g(20), g(8))
        grid([-1500, 0, -700], [0, 0, 1400], [0, 1200, 0], g(10), g(8))
        grid([1500, 0, -700], [0, 1200, 0], [0, 0, 1400], g(8), g(10))
        for z in (-350.0, 350.0):
            for i in range(12):
                x = -1320.0 + i * 240.0

                def column(S, T, x=x, z=z):
                    ang = 2 * np.pi * S
                    return np.stack([x + 38.0 * np.cos(ang), 520.0 * T, z + 38.0 * np.sin(ang)], -1)
                grid([0, 0, 0], [0, 0, 0], [0, 0, 0], g(6), g(6), displace=column)
                lo, hi = np.array([x - 55, 520, z - 55]), np.array([x + 55, 560, z + 55])
                ex, ey, ez = np.diag(hi - lo)
                for o, du, dv in ((lo, ex, ey), (lo + ez, ey, ex), (lo, ey, ez), (lo + ex, ez, ey), (lo, ez, ex), (lo + ey, ex, ez)):
                    grid(o, du, dv, g(1), g(1))
                if i < 11:  # arch between two columns
                    def arch(S, T, x=x, z=z):
                        ang = np.pi * S
                        return np.stack([x + 120 - 120 * np.cos(ang), 560 + 90 * np.sin(ang), z - 40 + 80 * T], -1)
                    grid([0, 0, 0], [0, 0, 0], [0, 0, 0], g(6), g(2), displace=arch)
            grid([-1440, 650, z], [2880, 0, 0], [0, 200, 0], g(24), g(2))
        for i in range(6):  # drapes
            x = -1100.0 + i * 440.0
            ph = rs.uniform(0, 6.28)

            def drape(S, T, ph=ph):
                return np.stack([0 * S, 0 * S, 35 * np.sin(14 * S + ph) * (0.3 + T)], -1)
            grid([x, 980, -40], [260, 0, 0], [0, -620, 0], g(16), g(16), displace=drape)
        return count[0], parts

    k0 = np.sqrt(target_tris / 7488.0)  # the counts above sum to 7488 k^2 before rounding
    best_k = min((k0 * f for f in np.linspace(0.8, 1.25, 181)), key=lambda k: abs(build(k, True)[0] - target_tris))
    return _finish(np.concatenate(build(best_k, False)[1], axis=0))


def terrain_objects(target_tris=1 << 20, seed=3):
    """Coherent surfaces: a displaced height field in [-10,10]^2 plus spheres resting on it."""
    rs = np.random.RandomState(seed)
    n_obj = int(min(48, max(1, target_tris // 4000)))
    n_sph = max(3, int(np.sqrt(target_tris * 0.25 / n_obj / 4)))
    n_ter = max(4, int(np.sqrt(max(32, target_tris - n_obj * 4 * n_sph * n_sph) / 2)))

    def height(S, T):
        x, z = 20 * S - 10, 20 * T - 10
        y = 0.8 * np.sin(0.9 * x) * np.cos(0.7 * z) + 0.25 * np.sin(3.1 * x + 1.3) * np.sin(2.7 * z) \
            + 0.05 * np.sin(17 * x) * np.sin(19 * z)
        return np.stack([0 * S, y, 0 * S], -1)
    parts = [_quad_grid([-10, 0, -10], [20, 0, 0], [0, 0, 20], n_ter, n_ter, displace=height)]
    for _ in range(n_obj):
        c = rs.uniform(-9, 9, size=3)
        r = rs.uniform(0.3, 1.1)
        c[1] = 1.0 + r
        parts.append(_sphere(c, r, n_sph))
    return _finish(np.concatenate(parts, axis=0))


def skewed(n_tris=262144, seed=5):
    """'Teapot in a stadium': triangle centres scaled by r^4 and log-normal triangle sizes, which gives
    spatial-median and SAH builders very unbalanced (deep) trees."""
    rs = np.random.RandomState(seed)
    c = rs.uniform(-1, 1, size=(n_tris, 3))
    r = np.linalg.norm(c, axis=1, keepdims=True)
    c = 10.0 * c * r ** 3
    size = 0.02 * np.exp(rs.normal(0.0, 1.0, size=(n_tris, 1, 1)))
    off = rs.uniform(-1, 1, size=(n_tris, 3, 3)) * size * (0.05 + r[:, None] ** 2)
    return _finish((c[:, None, :] + off).astype(np.float32))


def random_rays(n, seed=7, extent=10.0):
    """Uniformly random origins in the cube and uniformly random unit directions: (n, 8) float32 rows
    o.xyz, d.xyz, t_min = 1e-4, t_max = FLT_MAX."""
    rs = np.random.RandomState(seed)
    o = rs.uniform(-extent, extent, size=(n, 3))
    d = rs.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.empty((n, 8), np.float32)
    rays[:, 0:3] = o
    rays[:, 3:6] = d
    rays[:, 6] = 1e-4
    rays[:, 7] = np.finfo(np.float32).max
    return rays


def stadium_sheets(n_tris=1 << 22, sheets=40, seed=11):
    """Deep-stack scene for spatial-median trees: `sheets` parallel sheets facing +x at x = -2^i, each twice as large
    as the one before and tessellated into n_tris / sheets jittered triangles. A spatial-median split along x peels the
    farthest sheet off at every level, so the tree is a chain of `sheets` levels with a full subtree hanging off each;
    a ray travelling along -x enters both children at every level of the chain and leaves the farther one pending:
    the traversal stack holds one entry per sheet still ahead, far more than fit in the LDS part of the stack (and than
    the 25 slots of the reference's own fixed stack, rt/bbvh-base/bvh2.cpp:419)."""
    rs = np.random.RandomState(seed)
    k = max(2, int(np.sqrt(n_tris / sheets / 2)))
    parts = []
    for i in range(sheets):
        x, s = -(2.0 ** i), 0.25 * 2.0 ** i
        jitter = rs.uniform(-0.2, 0.2, size=(k + 1, k + 1, 1)) * (2.0 * s / k)

        def wobble(S, T, jitter=jitter, s=s):
            # (in-plane jitter of the grid points: unequal centres, no exact ties for the builder's partition)
            return np.concatenate([0 * jitter, jitter, np.roll(jitter, 1, axis=0)], axis=-1)
        parts.append(_quad_grid([x, -s, -s], [0, 2 * s, 0], [0, 0, 2 * s], k, k, displace=wobble))
    return _finish(np.concatenate(parts, axis=0))


def stadium_rays(n, seed=3, sheets=40, along=0.6):
    """Rays for stadium_sheets: a share `along` of them starts in front of the first sheet and runs along -x through
    all sheets (slightly spread, so that they cross each sheet inside its extent), the rest starts between the sheets
    in random directions."""
    rs = np.random.RandomState(seed)
    rays = np.empty((n, 8), np.float32)
    n_along = int(n * along)
    o = np.zeros((n, 3))
    d = np.zeros((n, 3))
    o[:n_along, 0] = 2.0
    o[:n_along, 1:3] = rs.uniform(-0.2, 0.2, size=(n_along, 2))
    d[:n_along] = np.concatenate([-np.ones((n_along, 1)), rs.uniform(-0.2, 0.2, size=(n_along, 2))], axis=1)
    m = n - n_along
    level = rs.uniform(0, sheets - 1, size=m)
    o[n_along:, 0] = -(2.0 ** level)
    o[n_along:, 1:3] = rs.uniform(-0.2, 0.2, size=(m, 2)) * (2.0 ** level)[:, None]
    d[n_along:] = rs.normal(size=(m, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 0:3], rays[:, 3:6] = o, d
    rays[:, 6], rays[:, 7] = 1e-4, np.finfo(np.float32).max
    return rays
