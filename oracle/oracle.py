"""ctypes binding of the CPU oracle (oracle/rtgi_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py.  Nothing under raytracing_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librtgi_oracle.so")

HIT_DTYPE = np.dtype([("t", "<f4"), ("beta", "<f4"), ("gamma", "<f4"), ("ref", "<u4")])
NODE56_DTYPE = np.dtype([("box_l", "<f4", (6,)), ("box_r", "<f4", (6,)), ("link_l", "<i4"), ("link_r", "<i4")])
assert NODE56_DTYPE.itemsize == 56

SPLIT = {"om": 0, "sm": 1, "sah": 2}


def build_library(force=False):
    """Compile the oracle with g++ (needs only the C++ standard library)."""
    src = os.path.join(_HERE, "rtgi_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build_library()
        L = C.CDLL(_SO)
        vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int, C.c_float
        L.orc_tri_intersect.argtypes = [vp, vp, vp]
        L.orc_box_intersect.argtypes = [vp, vp, i32, vp]
        L.orc_cam_rays.argtypes = [vp, f32, i32, i32, vp, i32, i32, i32, vp]
        L.orc_scene_create.argtypes = [vp, u32, vp, u32]
        L.orc_scene_create.restype = vp
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_seq_closest_hit.argtypes = [vp, vp, u32, vp, i32]
        L.orc_seq_any_hit.argtypes = [vp, vp, u32, vp, i32]
        L.orc_naive_build.argtypes = [vp, u32, vp, u32]
        L.orc_naive_build.restype = vp
        L.orc_naive_destroy.argtypes = [vp]
        L.orc_naive_copy_triangles.argtypes = [vp, vp]
        L.orc_naive_closest_hit.argtypes = [vp, vp, u32, vp, i32]
        L.orc_naive_any_hit.argtypes = [vp, vp, u32, vp, i32]
        L.orc_bbvh_build.argtypes = [vp, u32, vp, u32, i32, i32, i32, i32, i32]
        L.orc_bbvh_build.restype = vp
        L.orc_bbvh_destroy.argtypes = [vp]
        for name in ("num_nodes", "root", "num_index", "num_triangles", "depth"):
            f = getattr(L, "orc_bbvh_" + name)
            f.argtypes = [vp]
            f.restype = u32
        L.orc_bbvh_copy_nodes.argtypes = [vp, vp]
        L.orc_bbvh_copy_index.argtypes = [vp, vp]
        L.orc_bbvh_copy_triangles.argtypes = [vp, vp]
        L.orc_bbvh_closest_hit.argtypes = [vp, vp, u32, vp, vp, vp, i32]
        L.orc_bbvh_any_hit.argtypes = [vp, vp, u32, vp, vp, i32]
        L.orc_max_threads.restype = i32
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def max_threads():
    return lib().orc_max_threads()


def tri_intersect(a, b, c, ray8):
    """libgi/intersect.h:19-69 on one triangle. Returns (hit, t, beta, gamma)."""
    abc = _f32(np.concatenate([a, b, c]))
    r = _f32(ray8)
    out = np.zeros(3, np.float32)
    ok = lib().orc_tri_intersect(_ptr(abc), _ptr(r), _ptr(out))
    return bool(ok), float(out[0]), float(out[1]), float(out[2])


def box_intersect(lo, hi, ray8, variant=4):
    """variant 0: libgi/intersect.h:71-137, variant 4: :204-236. Returns (hit, entry distance)."""
    b = _f32(np.concatenate([lo, hi]))
    r = _f32(ray8)
    e = np.zeros(1, np.float32)
    ok = lib().orc_box_intersect(_ptr(b), _ptr(r), variant, _ptr(e))
    return bool(ok), float(e[0])


def cam_rays(cam_dir, cam_pos, cam_up, fovy, w, h, offsets=None, stride=1):
    """libgi/camera.cpp:11-22 for every stride-th pixel, y-major. Returns (n, 8) float32."""
    cam = _f32(np.concatenate([cam_dir, cam_pos, cam_up]))
    ny, nx = len(range(0, h, stride)), len(range(0, w, stride))
    out = np.zeros((ny * nx, 8), np.float32)
    off = _f32(offsets) if offsets is not None else None
    lib().orc_cam_rays(_ptr(cam), fovy, w, h, _ptr(off), 0, 0, stride, _ptr(out))
    return out


class Seq:
    """rt/seq/seq.cpp brute force."""

    def __init__(self, pos, tri):
        self.pos, self.tri = _f32(pos), _u32(tri)
        self.h = lib().orc_scene_create(_ptr(self.pos), len(self.pos), _ptr(self.tri), len(self.tri))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def closest_hit(self, rays, threads=0):
        rays = _f32(rays)
        out = np.zeros(len(rays), HIT_DTYPE)
        lib().orc_seq_closest_hit(self.h, _ptr(rays), len(rays), _ptr(out), threads)
        return out

    def any_hit(self, rays, threads=0):
        rays = _f32(rays)
        out = np.zeros(len(rays), np.uint8)
        lib().orc_seq_any_hit(self.h, _ptr(rays), len(rays), _ptr(out), threads)
        return out


class NaiveBvh:
    """rt/bbvh-base/bvh.cpp naive_bvh; hit refs index the triangle list returned by triangles()."""

    def __init__(self, pos, tri):
        self.pos, tri = _f32(pos), _u32(tri)
        self.nt = len(tri)
        self.h = lib().orc_naive_build(_ptr(self.pos), len(self.pos), _ptr(tri), len(tri))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_naive_destroy(self.h)
            self.h = None

    def triangles(self):
        out = np.zeros((self.nt, 3), np.uint32)
        lib().orc_naive_copy_triangles(self.h, _ptr(out))
        return out

    def closest_hit(self, rays, threads=0):
        rays = _f32(rays)
        out = np.zeros(len(rays), HIT_DTYPE)
        lib().orc_naive_closest_hit(self.h, _ptr(rays), len(rays), _ptr(out), threads)
        return out

    def any_hit(self, rays, threads=0):
        rays = _f32(rays)
        out = np.zeros(len(rays), np.uint8)
        lib().orc_naive_any_hit(self.h, _ptr(rays), len(rays), _ptr(out), threads)
        return out


class Bbvh:
    """rt/bbvh-base/bvh2.cpp binary_bvh_tracer<layout, esc>."""

    def __init__(self, pos, tri, layout="flat", esc=False, split="om", planes=0, max_tris=1):
        self.pos, tri = _f32(pos), _u32(tri)
        self.layout = layout
        self.h = lib().orc_bbvh_build(_ptr(self.pos), len(self.pos), _ptr(tri), len(tri),
                                      0 if layout == "flat" else 1, int(esc), SPLIT[split], planes, max_tris)
        if not self.h:
            raise ValueError("oracle: build refused (empty scene, 'sah' without planes, or degenerate ESC input)")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_bbvh_destroy(self.h)
            self.h = None

    @property
    def root(self):
        return lib().orc_bbvh_root(self.h)

    @property
    def depth(self):
        return lib().orc_bbvh_depth(self.h)

    def nodes(self):
        out = np.zeros(lib().orc_bbvh_num_nodes(self.h), NODE56_DTYPE)
        lib().orc_bbvh_copy_nodes(self.h, _ptr(out))
        return out

    def index(self):
        out = np.zeros(lib().orc_bbvh_num_index(self.h), np.uint32)
        if len(out):
            lib().orc_bbvh_copy_index(self.h, _ptr(out))
        return out

    def triangles(self):
        out = np.zeros((lib().orc_bbvh_num_triangles(self.h), 3), np.uint32)
        lib().orc_bbvh_copy_triangles(self.h, _ptr(out))
        return out

    def closest_hit(self, rays, threads=0, counts=False, max_stack=False):
        rays = _f32(rays)
        out = np.zeros(len(rays), HIT_DTYPE)
        cnt = np.zeros((len(rays), 3), np.uint32) if counts else None
        ms = np.zeros(len(rays), np.uint32) if max_stack else None
        lib().orc_bbvh_closest_hit(self.h, _ptr(rays), len(rays), _ptr(out), _ptr(cnt), _ptr(ms), threads)
        res = (out,)
        if counts:
            res += (cnt,)
        if max_stack:
            res += (ms,)
        return res[0] if len(res) == 1 else res

    def any_hit(self, rays, threads=0, counts=False):
        rays = _f32(rays)
        out = np.zeros(len(rays), np.uint8)
        cnt = np.zeros((len(rays), 3), np.uint32) if counts else None
        lib().orc_bbvh_any_hit(self.h, _ptr(rays), len(rays), _ptr(out), _ptr(cnt), threads)
        return (out, cnt) if counts else out
