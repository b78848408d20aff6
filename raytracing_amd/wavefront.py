"""Ray populations around the tracer: camera rays, shadow rays and diffuse bounce rays.

These are the callers' side of the hot path (gi/primary-hit.cpp:16-56 and libgi/wavefront-rt.h:90-102 of
the reference tree "Assignment 3/rtgi-2021-a03-ref/"): they decide which rays the tracer sees. The reference
has no path tracer, so the bounce rays follow this module's own definition (cosine-weighted about the
normal that faces the incoming ray, counter-based random numbers); host implementation in numpy, used to
prepare workloads and as the definition the device kernels are checked against.
"""
import numpy as np

FLT_MAX = np.float32(np.finfo(np.float32).max)
EPS = np.float32(1e-4)  # ray::eps, libgi/rt.h:21


def camera_rays(cam_dir, cam_pos, cam_up, fovy, w, h, offsets=None):
    """cam_ray of libgi/camera.cpp:11-22 for every pixel, y-major, in float32 like the reference.
    offsets: optional (h, w, 2) sub-pixel jitter in [-0.5, 0.5). Returns (w*h, 8) float32 rays."""
    f = np.float32
    d, up = np.asarray(cam_dir, f), np.asarray(cam_up, f)
    near_h = f(np.tan(f(np.pi) * f(fovy) * f(0.5) / f(180.0)))
    near_w = f(f(w) / f(h)) * near_h

    def cross(a, b):
        return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], f)
    U = cross(d, up)
    V = cross(U, d)
    xs = np.arange(w, dtype=f)[None, :] + f(0.5)
    ys = np.arange(h, dtype=f)[:, None] + f(0.5)
    if offsets is not None:
        xs = xs + np.asarray(offsets, f)[..., 0]
        ys = ys + np.asarray(offsets, f)[..., 1]
    u = (near_w * (xs / f(w) * f(2.0) - f(1.0))).astype(f) + np.zeros((h, w), f)
    v = (near_h * (ys / f(h) * f(2.0) - f(1.0))).astype(f) + np.zeros((h, w), f)
    dirs = d[None, None, :] + U[None, None, :] * u[..., None] + V[None, None, :] * v[..., None]
    dirs = dirs.astype(f)
    sq = (dirs[..., 0] * dirs[..., 0] + dirs[..., 1] * dirs[..., 1]) + dirs[..., 2] * dirs[..., 2]
    dirs = dirs * (f(1.0) / np.sqrt(sq, dtype=f))[..., None]
    rays = np.empty((h * w, 8), f)
    rays[:, 0:3] = np.asarray(cam_pos, f)
    rays[:, 3:6] = dirs.reshape(-1, 3)
    rays[:, 6] = EPS
    rays[:, 7] = FLT_MAX
    return rays


def hit_points(pos, tri, hits):
    """diff_geom::x of libgi/rt.cpp:9: (1 - beta - gamma) a + beta b + gamma c, float32."""
    f = np.float32
    v = pos[tri[hits["ref"]]]
    b, g = hits["beta"][:, None], hits["gamma"][:, None]
    return ((f(1.0) - b - g) * v[:, 0] + b * v[:, 1] + g * v[:, 2]).astype(f)


def shadow_rays(points, light_pos):
    """Shadow rays of gi/primary-hit.cpp:43-50: from the hit point towards a point light, ending EPS
    before it (ray::length_exclusive, libgi/rt.h:26)."""
    f = np.float32
    to_light = (np.asarray(light_pos, f)[None, :] - points).astype(f)
    sq = (to_light[:, 0] * to_light[:, 0] + to_light[:, 1] * to_light[:, 1]) + to_light[:, 2] * to_light[:, 2]
    dist = np.sqrt(sq, dtype=f)
    rays = np.empty((len(points), 8), f)
    rays[:, 0:3] = points
    rays[:, 3:6] = to_light * (f(1.0) / dist)[:, None]
    rays[:, 6] = EPS
    rays[:, 7] = dist - EPS
    return rays


def _hash_u32(x):
    """Counter-based generator: a 32-bit integer finaliser (two multiply-xorshift rounds)."""
    x = np.asarray(x, np.uint64) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def uniform01(index, stream, seed=0):
    """Uniform floats in [0, 1) keyed by (ray index, stream number, seed); 24 random bits each."""
    key = _hash_u32(np.asarray(index, np.uint64) * np.uint64(0x9E3779B1) + np.uint64(stream) * np.uint64(0x85EBCA77)
                    + np.uint64(seed) * np.uint64(0xC2B2AE3D))
    key = _hash_u32(key ^ np.uint32(0x27D4EB2F))
    return (key >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / (1 << 24))


def bounce_rays(pos, tri, rays, hits, seed=0, bounce=1, ray_index=None):
    """One diffuse bounce: for every ray that hit something, a new ray from the hit point into a
    cosine-weighted random direction about the geometric normal that faces the incoming ray.
    Returns (new rays (m, 8) float32, indices of the parent rays (m,))."""
    f = np.float32
    valid = np.flatnonzero(hits["t"] != FLT_MAX)
    hv = hits[valid]
    x = hit_points(pos, tri, hv)
    v = pos[tri[hv["ref"]]].astype(f)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]).astype(f)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), f(1e-30))
    d_in = rays[valid, 3:6]
    flip = (n * d_in).sum(1) > 0
    n[flip] = -n[flip]
    idx = valid if ray_index is None else np.asarray(ray_index)[valid]
    u1 = uniform01(idx, 2 * bounce, seed)
    u2 = uniform01(idx, 2 * bounce + 1, seed)
    r, phi = np.sqrt(u1, dtype=f), f(2 * np.pi) * u2
    lx, ly = r * np.cos(phi), r * np.sin(phi)
    lz = np.sqrt(np.maximum(f(0), f(1) - u1), dtype=f)
    # orthonormal basis around n (branchless form of Duff et al.)
    sign = np.where(n[:, 2] >= 0, f(1), f(-1))
    a = f(-1) / (sign + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    t1 = np.stack([f(1) + sign * n[:, 0] * n[:, 0] * a, sign * b, -sign * n[:, 0]], 1)
    t2 = np.stack([b, sign + n[:, 1] * n[:, 1] * a, -n[:, 1]], 1)
    d = (t1 * lx[:, None] + t2 * ly[:, None] + n * lz[:, None]).astype(f)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out = np.empty((len(valid), 8), f)
    out[:, 0:3] = x
    out[:, 3:6] = d
    out[:, 6] = EPS
    out[:, 7] = FLT_MAX
    return out, valid


PI = np.float32(np.pi)                # libgi/rt.h:14
ONE_OVER_PI = np.float32(1.0 / np.pi)  # libgi/rt.h:15


def shading_normals(normals, tri):
    """diff_geom::ns of libgi/rt.cpp:10: normalize(a.norm + b.norm + c.norm) per triangle, float32."""
    f = np.float32
    n = np.asarray(normals, f)[tri]
    s = ((n[:, 0] + n[:, 1]) + n[:, 2]).astype(f)
    sq = (s[:, 0] * s[:, 0] + s[:, 1] * s[:, 1]) + s[:, 2] * s[:, 2]
    return (s * (f(1.0) / np.sqrt(sq, dtype=f))[:, None]).astype(f)


def texture_albedo(tex_coords, tri, hits, texture):
    """diff_geom::albedo through a texture (libgi/rt.cpp:11, 27-31; texture::sample, libgi/scene.h:24-32): texture
    coordinate interpolated with the hit's barycentrics, wrapped, nearest texel. tri: the triangles hit ids index;
    texture: (h, w, 3) float32. Returns (n, 3) for the hits given (all must be hits)."""
    f = np.float32
    tc = np.asarray(tex_coords, f)[tri[hits["ref"]]]  # (n, 3, 2)
    beta, gamma = hits["beta"].astype(f), hits["gamma"].astype(f)
    alpha = ((f(1.0) - beta) - gamma).astype(f)
    uv = ((alpha[:, None] * tc[:, 0] + beta[:, None] * tc[:, 1]).astype(f) + gamma[:, None] * tc[:, 2]).astype(f)
    uv = (uv - np.floor(uv)).astype(f)
    th, tw = texture.shape[0], texture.shape[1]
    x = (uv[:, 0] * f(tw) + f(0.5)).astype(f).astype(np.int64)
    y = (uv[:, 1] * f(th) + f(0.5)).astype(f).astype(np.int64)
    x[x == tw] = 0
    y[y == th] = 0
    return np.asarray(texture, f)[y, x]


def albedo_of_hits(hits, material_id, albedo, tex_coords=None, tri=None, textures=None, material_texture=None):
    """Albedo per valid hit: the material's colour, or its texture's texel where it has one."""
    valid = hits["t"] != FLT_MAX
    hv = hits[valid]
    mid = np.asarray(material_id)[hv["ref"]]
    out = np.asarray(albedo, np.float32)[mid].copy()
    if textures:
        mt = np.asarray(material_texture)[mid]
        for k, tex in enumerate(textures):
            sel = mt == k
            if sel.any():
                out[sel] = texture_albedo(tex_coords, tri, hv[sel], tex)
    return valid, out


def shade_primary(hits, material_id, albedo):
    """primary_hit_display::sample_pixel, gi/primary-hit.cpp:16-30: albedo of the hit triangle, black for a miss."""
    out = np.zeros((len(hits), 4), np.float32)
    valid = hits["t"] != FLT_MAX
    out[valid, 0:3] = np.asarray(albedo, np.float32)[np.asarray(material_id)[hits["ref"][valid]]]
    return out


def shade_local(pos, tri, ns, material_id, albedo, hits, occluded, light_pos, light_col, albedo_fn=None):
    """local_illumination::sample_pixel, gi/primary-hit.cpp:32-57, with the lambertian BRDF
    (libgi/material.cpp:17-20) and pointlight::power = 4 pi col (libgi/scene.cpp:216-218)."""
    f = np.float32
    out = np.zeros((len(hits), 4), f)
    lit = np.flatnonzero((hits["t"] != FLT_MAX) & (np.asarray(occluded) == 0))
    hv = hits[lit]
    x = hit_points(pos, tri, hv)
    to_light = (np.asarray(light_pos, f)[None, :] - x).astype(f)
    sq = (to_light[:, 0] * to_light[:, 0] + to_light[:, 1] * to_light[:, 1]) + to_light[:, 2] * to_light[:, 2]
    d = np.sqrt(sq, dtype=f)
    w_i = to_light * (f(1.0) / d)[:, None]
    n = ns[hv["ref"]]
    facing = ((n[:, 0] * w_i[:, 0] + n[:, 1] * w_i[:, 1]) + n[:, 2] * w_i[:, 2]) > 0
    power = (f(4) * PI) * np.asarray(light_col, f)
    alb = np.asarray(albedo, f)[np.asarray(material_id)[hv["ref"]]] if albedo_fn is None else albedo_fn(hv)
    rad = (power[None, :] * (ONE_OVER_PI * alb)) / (d * d)[:, None]
    rad[~facing] = 0
    out[lit, 0:3] = rad
    return out


def _dot(a, b):
    return ((a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]).astype(np.float32)


def _cdot(a, b):
    x = _dot(a, b)
    return np.where(x < 0, np.float32(0), x).astype(np.float32)


def fresnel_dielectric(cos_wi, ior_medium, ior_material):
    """libgi/util.h:44-60, float32, elementwise."""
    f = np.float32
    with np.errstate(all="ignore"):
        ei = np.where(cos_wi < 0, ior_material, ior_medium).astype(f)
        et = np.where(cos_wi < 0, ior_medium, ior_material).astype(f)
        c = np.minimum(np.maximum(np.abs(cos_wi), f(0)), f(1)).astype(f)
        sin_t = ((ei / et) * np.sqrt(f(1) - c * c, dtype=f)).astype(f)
        rev = (f(1) - sin_t * sin_t).astype(f)
        cos_t = np.sqrt(np.where(rev > 0, rev, f(0)).astype(f), dtype=f)
        rparl = ((et * c) - (ei * cos_t)) / ((et * c) + (ei * cos_t))
        rperp = ((ei * c) - (et * cos_t)) / ((ei * c) + (et * cos_t))
        out = ((rparl * rparl + rperp * rperp) / f(2)).astype(f)
    return np.where(sin_t >= 1, f(1), out).astype(f)


def _tan2(c):
    f = np.float32
    with np.errstate(all="ignore"):
        res = (f(1) - c * c).astype(f)
        return (np.where(res < 0, f(0), res) / (c * c)).astype(f)


def _ggx_d(n_h, roughness):
    f = np.float32
    with np.errstate(all="ignore"):
        tan2 = _tan2(n_h)
        a2 = (roughness * roughness).astype(f)
        n2, s = (n_h * n_h).astype(f), (a2 + tan2).astype(f)
        v = (a2 / ((PI * (n2 * n2)) * (s * s))).astype(f)
    return np.where((n_h <= 0) | ~np.isfinite(tan2), f(0), v).astype(f)


def _ggx_g1(n_v, roughness):
    f = np.float32
    with np.errstate(all="ignore"):
        tan2 = _tan2(n_v)
        v = (f(2) / (f(1) + np.sqrt(f(1) + (roughness * roughness) * tan2, dtype=f))).astype(f)
    return np.where((n_v <= 0) | ~np.isfinite(tan2), f(0), v).astype(f)


BRDF_NAMES = ("lambert", "phong", "layered-phong", "gtr2", "layered-gtr2")  # new_brdf, libgi/material.cpp:77-102


def brdf_f(kind, n, w_o, w_i, alb, ior, roughness):
    """brdf::f of libgi/material.cpp:7-75 for arrays of shading points (ng = ns, libgi/rt.cpp:13), float32 with the
    reference's operation order; kind: index into BRDF_NAMES per point."""
    f = np.float32
    m = len(n)
    facing = _dot(n, w_i) > 0
    lam = np.where(facing[:, None], ONE_OVER_PI * alb, f(0)).astype(f)
    with np.errstate(all="ignore"):
        # phong (material.cpp:25-32)
        exponent = (f(2) / (roughness * roughness) - f(2)).astype(f)
        d = _dot(w_i, n)
        r = ((f(2) * n) * d[:, None] - w_i).astype(f)
        cos_theta = _cdot(w_o, r)
        norm_f = ((exponent + f(2)) * np.float32(1.0 / (2 * np.pi))).astype(f)
        p = np.power(cos_theta, exponent, dtype=f)
        c = _cdot(w_i, n)

        def phong(col):
            return np.where(facing[:, None], ((col * p[:, None]) * norm_f[:, None]) * c[:, None], f(0)).astype(f)
        # gtr2 (material.cpp:58-73)
        n_v, n_l = _cdot(n, w_o), _cdot(n, w_i)
        h = (w_o + w_i).astype(f)
        H = (h * (f(1) / np.sqrt(_dot(h, h), dtype=f))[:, None]).astype(f)
        n_h, h_l = _cdot(n, H), _cdot(H, w_i)
        F = fresnel_dielectric(h_l, np.full(m, 1, f), ior)
        D = _ggx_d(n_h, roughness)
        G = (_ggx_g1(n_v, roughness) * _ggx_g1(n_l, roughness)).astype(f)
        micro = (((F * D) * G) / ((f(4) * np.abs(n_v)) * np.abs(n_l))).astype(f)
        micro = np.where(facing & (n_v != 0), micro, f(0)).astype(f)

        def gtr2(coat):
            return (np.repeat(micro[:, None], 3, 1) if coat else alb * micro[:, None]).astype(f)
        one = np.ones((m, 3), f)
        F_coat = fresnel_dielectric(np.abs(_dot(n, w_o)), np.full(m, 1, f), ior)[:, None]
        variants = [lam, phong(alb), (f(1) - F_coat) * lam + F_coat * phong(one), gtr2(False),
                    (f(1) - F_coat) * lam + F_coat * gtr2(True)]
    out = np.zeros((m, 3), f)
    for k, v in enumerate(variants):
        sel = np.asarray(kind) == k
        out[sel] = v[sel]
    return out


def shade_local_brdf(pos, tri, ns, material_id, materials, rays, hits, occluded, light_pos, light_col, albedo_fn=None):
    """shade_local with each hit material's own BRDF. materials: dict(albedo (n, 3), ior, roughness, kind); rays: the
    (n, 8) rays of the hits (w_o = -d, gi/primary-hit.cpp:45)."""
    f = np.float32
    out = np.zeros((len(hits), 4), f)
    lit = np.flatnonzero((hits["t"] != FLT_MAX) & (np.asarray(occluded) == 0))
    hv = hits[lit]
    x = hit_points(pos, tri, hv)
    to_light = (np.asarray(light_pos, f)[None, :] - x).astype(f)
    d = np.sqrt(_dot(to_light, to_light), dtype=f)
    w_i = (to_light * (f(1.0) / d)[:, None]).astype(f)
    w_o = (-np.asarray(rays, f)[lit, 3:6]).astype(f)
    mid = np.asarray(material_id)[hv["ref"]]
    alb = np.asarray(materials["albedo"], f)[mid] if albedo_fn is None else albedo_fn(hv)
    fr = brdf_f(np.asarray(materials["kind"])[mid], ns[hv["ref"]], w_o, w_i, alb,
                np.asarray(materials["ior"], f)[mid], np.asarray(materials["roughness"], f)[mid])
    power = (f(4) * PI) * np.asarray(light_col, f)
    with np.errstate(all="ignore"):
        out[lit, 0:3] = (power[None, :] * fr) / (d * d)[:, None]
    return out


def framebuffer_add(fb, radiance):
    """framebuffer::add, libgi/framebuffer.cpp:10-18, one sample per pixel: running mean in rgb, count in w."""
    f = np.float32
    new_count = (fb[:, 3] + f(1.0)).astype(f)
    out = np.empty_like(fb)
    out[:, 0:3] = (fb[:, 0:3] * fb[:, 3:4] + radiance[:, 0:3]) / new_count[:, None]
    out[:, 3] = new_count
    return out


def framebuffer_rgb8(fb, w, h):
    """framebuffer::png, libgi/framebuffer.cpp:21-30: clamp, gamma 1/2.2, * 255, truncate, rows flipped."""
    f = np.float32
    c = np.clip(fb[:, 0:3].reshape(h, w, 3), f(0), f(1))
    c = np.power(c, f(1.0 / 2.2), dtype=f) * f(255.0)
    return c.astype(np.uint8)[::-1].copy()
