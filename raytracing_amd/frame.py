"""ctypes binding of librtgi_hip_frame.so (include/rtgi_hip_frame.h): whole frames of the wavefront algorithms on one
GPU or several, host side in C++ (one thread and one stream per GPU, RCCL for the framebuffer exchange). What it
stands for in the reference tree "Assignment 3/rtgi-2021-a03-ref/": `raytracer ...` + `commit` + `run`
(driver/interaction.cpp:177-212, driver/main.cpp:38-48).

Used by the tests, bench.py and the examples; there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import tracer

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtgi_hip_frame.so")

ALGO = {"primary": 0, "local": 1, "path": 2}
SHARDING = {"tiles": 0, "samples": 1}

# every symbol include/rtgi_hip_frame.h declares
EXPORTS = [
    "rtgi_hip_renderer_create", "rtgi_hip_renderer_unique_id", "rtgi_hip_renderer_create_rank", "rtgi_hip_renderer_destroy", "rtgi_hip_renderer_device_count", "rtgi_hip_renderer_handle",
    "rtgi_hip_renderer_interprete", "rtgi_hip_renderer_build", "rtgi_hip_renderer_set_materials", "rtgi_hip_renderer_set_textures", "rtgi_hip_renderer_render", "rtgi_hip_renderer_rt_bench", "rtgi_hip_renderer_last_error",
    "rtgi_hip_frame_share",
]


def frame_share(sharding, w, h, tile, sppx, world, rank):
    """The pixels (sharding "tiles") or sample indices ("samples") that part `rank` of `world` takes of a frame, as the
    frame library deals them (host code, needs no GPU)."""
    import numpy as np
    L = lib()
    kind = {"tiles": 0, "samples": 1}[sharding]
    n = C.c_uint32(0)
    rc = L.rtgi_hip_frame_share(kind, w, h, tile, sppx, world, rank, None, C.byref(n))
    if rc:
        raise ValueError("rtgi_hip_frame_share refused its arguments (%d)" % rc)
    out = np.empty(n.value, np.uint32)
    if n.value:
        L.rtgi_hip_frame_share(kind, w, h, tile, sppx, world, rank, out.ctypes.data, C.byref(n))
    return out


class Camera(C.Structure):
    _fields_ = [("dir", C.c_float * 3), ("pos", C.c_float * 3), ("up", C.c_float * 3), ("fovy", C.c_float),
                ("w", C.c_int32), ("h", C.c_int32)]


class Frame(C.Structure):
    _fields_ = [("cam", Camera), ("sppx", C.c_uint32), ("algo", C.c_uint32), ("bounces", C.c_uint32),
                ("light_pos", C.c_float * 3), ("light_col", C.c_float * 3), ("seed", C.c_uint32), ("jitter", C.c_uint32),
                ("sharding", C.c_uint32), ("tile", C.c_uint32)]


class FrameStats(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_trace", C.c_double), ("ms_reduce", C.c_double), ("rays", C.c_uint64),
                ("n_devices", C.c_int32), ("used_rccl", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        tracer.lib()  # (also makes torch's copy of the HIP runtime the one in the process)
        if not os.path.exists(LIB_PATH):
            raise tracer.RtgiHipError("%s is missing: build it with `make -C raytracing_amd/csrc all`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32, u32, sz, cp = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t, C.c_char_p
        L.rtgi_hip_renderer_create.argtypes = [C.POINTER(C.c_int), i32, i32, i32]
        L.rtgi_hip_renderer_create.restype = vp
        L.rtgi_hip_renderer_unique_id.argtypes = [vp, sz]
        L.rtgi_hip_renderer_create_rank.argtypes = [i32, i32, i32, i32, i32, vp]
        L.rtgi_hip_renderer_create_rank.restype = vp
        L.rtgi_hip_renderer_destroy.argtypes = [vp]
        L.rtgi_hip_renderer_destroy.restype = None
        L.rtgi_hip_renderer_device_count.argtypes = [vp]
        L.rtgi_hip_renderer_handle.argtypes = [vp, i32]
        L.rtgi_hip_renderer_handle.restype = vp
        L.rtgi_hip_renderer_interprete.argtypes = [vp, cp, cp]
        L.rtgi_hip_renderer_build.argtypes = [vp, vp, sz, u32, vp, sz, u32, vp, u32]
        L.rtgi_hip_renderer_set_materials.argtypes = [vp, vp, u32]
        L.rtgi_hip_renderer_set_textures.argtypes = [vp, vp, sz, u32, vp, sz, u32, vp, u32, vp, u32]
        L.rtgi_hip_renderer_render.argtypes = [vp, C.POINTER(Frame), vp, vp, C.POINTER(FrameStats)]
        L.rtgi_hip_renderer_rt_bench.argtypes = [vp, C.POINTER(Frame), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.rtgi_hip_renderer_last_error.argtypes = [vp]
        L.rtgi_hip_frame_share.argtypes = [u32, i32, i32, u32, u32, i32, i32, vp, C.POINTER(u32)]
        L.rtgi_hip_renderer_last_error.restype = cp
        _lib = L
    return _lib


class HipRenderer:
    """`raytracer hip [indexed] [esc] devices N` with the frame loop of `run` on the GPUs."""

    def __init__(self, devices=(0,), layout="flat", esc=False, rank=None, world=None, unique_id=None):
        """devices: the GPUs of this process. With rank/world/unique_id (one process per GPU): this process's GPU is
        devices[0], unique_id the bytes of unique_id() as handed out by one of the processes."""
        if rank is None:
            devs = (C.c_int * len(devices))(*devices)
            self._r = lib().rtgi_hip_renderer_create(devs, len(devices), tracer.LAYOUT[layout], int(esc))
        else:
            buf = C.create_string_buffer(bytes(unique_id), 128)
            self._r = lib().rtgi_hip_renderer_create_rank(devices[0], tracer.LAYOUT[layout], int(esc), rank, world, buf)
        if not self._r:
            raise tracer.RtgiHipError(lib().rtgi_hip_renderer_last_error(None).decode())
        self.devices = tuple(devices)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        if lib().rtgi_hip_renderer_unique_id(buf, 128) < 0:
            raise tracer.RtgiHipError(lib().rtgi_hip_renderer_last_error(None).decode())
        return buf.raw

    def close(self):
        if getattr(self, "_r", None):
            lib().rtgi_hip_renderer_destroy(self._r)
            self._r = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise tracer.RtgiHipError("%s (code %d)" % (lib().rtgi_hip_renderer_last_error(self._r).decode(), rc))
        return rc

    def interprete(self, command, args=""):
        return self._check(lib().rtgi_hip_renderer_interprete(self._r, command.encode(), args.encode())) == 1

    def build(self, pos, normals, tri, material_id, albedo):
        """The scene as the renderer holds it: 32-byte vertex records (position, normal, texture coordinate) and
        16-byte triangle records (a, b, c, material id), libgi/rt.h:29-38."""
        nv, nt = len(pos), len(tri)
        vertices = np.zeros((nv, 8), np.float32)
        vertices[:, 0:3], vertices[:, 3:6] = pos, normals
        tri4 = np.empty((nt, 4), np.uint32)
        tri4[:, 0:3], tri4[:, 3] = tri, material_id
        albedo = np.ascontiguousarray(albedo, dtype=np.float32)
        self._check(lib().rtgi_hip_renderer_build(self._r, vertices.ctypes.data, 32, nv, tri4.ctypes.data, 16, nt,
                                                  albedo.ctypes.data, len(albedo)))

    def set_materials(self, albedo, ior, roughness, brdf):
        rec = np.zeros(len(albedo), tracer.MATERIAL_DTYPE)
        rec["albedo"], rec["ior"], rec["roughness"] = albedo, ior, roughness
        rec["brdf"] = [tracer.BRDF[b] for b in brdf]
        self._check(lib().rtgi_hip_renderer_set_materials(self._r, rec.ctypes.data, len(rec)))

    def rt_bench(self, cam_dir, cam_pos, cam_up, fovy, w, h):
        """`rt_bench` (driver/main.cpp:50-65): closest_hit on one frame of camera rays -> (rays, ms)."""
        f3 = C.c_float * 3
        frame = Frame(Camera(f3(*cam_dir), f3(*cam_pos), f3(*cam_up), fovy, w, h), 1, 0, 0, f3(0, 0, 0), f3(1, 1, 1), 0, 0, 0, 0)
        n, ms = C.c_uint64(0), C.c_double(0.0)
        self._check(lib().rtgi_hip_renderer_rt_bench(self._r, C.byref(frame), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def render(self, cam_dir, cam_pos, cam_up, fovy, w, h, sppx=1, algo="local", bounces=0, light_pos=(0, 0, 0),
               light_col=(1, 1, 1), seed=0, jitter=True, sharding="tiles", tile=0, want_rgb8=True):
        """-> (framebuffer (w*h, 4) float32: mean rgb and sample count, rgb8 (h, w, 3) uint8 or None, stats dict)."""
        f3 = C.c_float * 3
        frame = Frame(Camera(f3(*cam_dir), f3(*cam_pos), f3(*cam_up), fovy, w, h), sppx, ALGO[algo], bounces, f3(*light_pos),
                      f3(*light_col), seed, int(bool(jitter)), SHARDING[sharding], tile)
        fb = np.empty((w * h, 4), np.float32)
        rgb = np.empty((h, w, 3), np.uint8) if want_rgb8 else None
        st = FrameStats()
        self._check(lib().rtgi_hip_renderer_render(self._r, C.byref(frame), fb.ctypes.data, rgb.ctypes.data if want_rgb8 else None,
                                                   C.byref(st)))
        stats = {k: getattr(st, k) for k, _ in FrameStats._fields_}
        return fb, rgb, stats
