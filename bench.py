#!/usr/bin/env python3
"""Ray-traversal throughput of the HIP backend: Mrays/s of closest_hit on incoherent first-bounce rays.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched through torch.distributed.run)

One step = one closest_hit pass over one batch of diffuse-bounce rays that already live in GPU memory
(workload "soup1m_bounce": 2^20-triangle seeded soup, 1920x1080 camera rays from the centre of the scene,
one cosine-weighted bounce off every primary hit). With N GPUs every rank holds the whole BVH and traces
its own sample's bounce batch (sharding by sample index: weak scaling, no collective on the data path).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, vendor peak
L2_PEAK_GBS = 34500.0  # aggregate of the eight 4 MiB L2s (the figure of the MI355X guide this repository is built against)
# HBM is not what the traversal runs against: nodes and triangles of the 2^20-triangle scene are 112 MiB and live in the
# 256 MiB Infinity Cache and mostly in the L2s. The line therefore reports, beside the reference's algorithmic bytes over
# the kernel time, (a) the bytes that the hardware counters saw beyond L2 and in front of it against the HBM and L2 peaks,
# (b) how busy the counters found the units of a CU's vector memory pipeline and its SIMDs, and (c) the time each resource
# would need for this launch's work alone. One of those resources is the rate at which a CU's vector memory pipeline takes
# 16-byte loads whose 64 lanes go to 64 different cache lines: scripts/ubench/gather_mix measures it with the kernel's
# access shape (dependent random 64-byte records, 77 % of them L2 hits, four waves per SIMD like the kernel); bench.py
# runs it on the box when the binary is there, else it uses the figure of profiles/r02_ubench_gather_mix_3waves.txt.
LANE_LOADS_PER_S_FALLBACK = 4 * 213.2e9
MISS_RECORDS_PER_S_FALLBACK = 71.6e9
SIMDS, VALU_CYCLES_PER_INSTRUCTION = 1024, 4

WORKLOADS = {
    # name: (scene generator, kwargs, camera position, direction, up)
    "soup1m_bounce": ("soup", dict(n_tris=1 << 20, seed=1), (0, 0, 0), (0, 0, -1), (0, 1, 0)),
    "soup260k_bounce": ("soup", dict(n_tris=1 << 18, seed=1), (0, 0, 0), (0, 0, -1), (0, 1, 0)),
    "terrain1m_bounce": ("terrain_objects", dict(target_tris=1 << 20, seed=3), (0, 6, 14), (0, -0.35, -0.937), (0, 1, 0)),
    "sponza260k_bounce": ("sponza_class", dict(target_tris=262144, seed=0), (-516, 300, 0), (1, 0, 0), (0, 1, 0)),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="soup1m_bounce", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--layout", default="flat", choices=["flat", "indexed"])
    ap.add_argument("--bvh", default="", help="extra 'bvh ...' commands separated by ';', e.g. 'sah 16'")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-host-buffers", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the primary-ray and shadow-ray rates")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the launch on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--frame", action="store_true",
                    help="other mode: whole path-traced frames through the C++ frame level (librtgi_hip_frame.so), tiles dealt "
                         "over the GPUs, framebuffer all-reduce (RCCL, called from C++) inside the timed region")
    ap.add_argument("--sppx", type=int, default=64, help="--frame: samples per pixel")
    ap.add_argument("--bounces", type=int, default=1, help="--frame: path segments after the camera ray")
    ap.add_argument("--sharding", default="tiles", choices=["tiles", "samples"], help="--frame: how a frame is dealt over the GPUs")
    ap.add_argument("--presort", default="none", choices=["none", "morton", "octant_morton", "morton_octant", "random"],
                    help="experiment: reorder the bounce batch on the host before the timed region")
    return ap.parse_args()


def build_workload(args, rank, device):
    """Scene, tracer and this rank's bounce batch. Everything here is outside the timed region."""
    import torch
    from raytracing_amd import scenes, tracer, wavefront
    gen, kw, cam_pos, cam_dir, cam_up = WORKLOADS[args.workload]
    pos, tri, _ = getattr(scenes, gen)(**kw)
    t = tracer.HipBvhTracer(layout=args.layout, device=device)
    for cmd in [c.strip() for c in args.bvh.split(";") if c.strip()]:
        t.interprete("bvh", cmd)
    t0 = time.time()
    t.build(pos, tri)
    build_s = time.time() - t0
    tri_original = tri
    if args.layout == "flat":  # commit_shuffled_triangles: hit ids index the permuted list
        tri = tri[t.triangle_order()]
    # sample index = rank: jittered camera rays, closest_hit, then one diffuse bounce -- all on the GPU
    dev = "cuda:%d" % device
    npix = args.width * args.height
    jitter = np.stack([wavefront.uniform01(np.arange(npix), 100, rank) - 0.5,
                       wavefront.uniform01(np.arange(npix), 101, rank) - 0.5], 1).astype(np.float32)
    cam_dir = np.asarray(cam_dir, np.float32) / np.float32(np.linalg.norm(cam_dir))
    d_primary = t.camera_rays_device(cam_dir, cam_pos, cam_up, 65.0, args.width, args.height,
                                     d_offsets=torch.from_numpy(jitter).to(dev))
    d_hits = torch.empty((npix, 4), dtype=torch.float32, device=dev)
    t.closest_hit_device(d_primary, d_hits)
    d_bounce, _ = t.bounce_rays_device(d_primary, d_hits, seed=rank, bounce=1)
    torch.cuda.synchronize(device)
    hit_rate = float((d_hits[:, 0] != tracer.FLT_MAX).float().mean().item())
    return dict(tracer=t, pos=pos, tri=tri, tri_original=tri_original, bounce=d_bounce.contiguous().cpu().numpy(),
                build_s=build_s, primary_hit_rate=hit_rate)


def algorithmic_bytes(t, d_rays, layout):
    """Bytes the reference's data structures would move for these rays: 32 B ray in, 16 B hit out, 56 B per
    inner node popped (sizeof(node), rt/bbvh-base/bvh.h:39-47), 52 B per triangle tested (16-B triangle
    record + three 12-B positions, libgi/intersect.h:20-33), 4 B more per test through index[] (bvh.h:102-106).
    The pops and tests are counted by the kernel itself; tests check them against the oracle's counts."""
    import torch
    n = d_rays.shape[0]
    d_hits = torch.empty((n, 4), dtype=torch.float32, device=d_rays.device)
    d_cnt = torch.zeros((n, 2), dtype=torch.int32, device=d_rays.device)
    t.profile_begin()
    t.closest_hit_device(d_rays, d_hits, d_cnt)
    torch.cuda.synchronize(d_rays.device)
    work = t.profile_work()  # rays, inner pops, triangle tests, inner pops served from LDS
    s = d_cnt.sum(0, dtype=torch.int64).cpu().numpy()
    assert work[0] == n and work[1] == int(s[0]) and work[2] == int(s[1])
    per_tri = 52 + (4 if layout == "indexed" else 0)
    total = 48 * n + 56 * int(s[0]) + per_tri * int(s[1])
    hit_rate = float((d_hits[:, 0] != float(np.finfo(np.float32).max)).float().mean().item())
    # 16-byte loads the lanes send to the vector cache: three per inner node not served from LDS (the twelve box
    # planes), one more where the node's links are not implied by its index, three per triangle, two per ray
    details = t.profile_details()
    lane_loads = 3 * (work[1] - work[3]) + details["link_fetches"] + 3 * work[2] + 2 * n
    t.profile_end()
    return dict(bytes=total, pops=float(s[0]) / n, tests=float(s[1]) / n, hit_rate=hit_rate, lane_loads=int(lane_loads),
                lds_share=work[3] / max(work[1], 1), max_pops=int(d_cnt[:, 0].max().item()), hits=d_hits,
                link_fetch_share=details["link_fetches"] / max(work[1], 1), link_fetches=int(details["link_fetches"]))


def cpu_baseline(args, w, gpu_hits):
    """The CPU oracle (restated reference traversal) on a bounded sample of the same rays: all host threads, then
    one thread on a sixteenth of the batch. Also the checker: its hit records must be the GPU's."""
    from oracle import oracle as orc
    # the oracle permutes the triangle list itself for the flat layout: hand it the original order
    o = orc.Bbvh(w["pos"], w["tri_original"], layout=args.layout, **_oracle_split(args.bvh))
    rays = w["bounce"]
    o.closest_hit(rays[:20000])  # warm-up: thread pool, caches
    passes, dt, t0 = 0, 0.0, time.time()
    while dt < args.cpu_seconds:  # whole passes over the batch until the time budget is used
        cpu_hits = o.closest_hit(rays)
        passes += 1
        dt = time.time() - t0
    identical = bool(cpu_hits.tobytes() == gpu_hits.tobytes())
    part = rays[::16]
    t1 = time.time()
    o.closest_hit(part, threads=1)
    dt1 = time.time() - t1
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    return {"value": round(passes * len(rays) / dt / 1e6, 4), "unit": "Mrays/s", "cores": orc.max_threads(), "kind": "port",
            "one_thread": round(len(part) / dt1 / 1e6, 4), "host_cpu": model, "host_logical_cpus": os.cpu_count(),
            "hits_identical_to_gpu": identical,
            "sample": "%d passes over the same %d bounce rays in %.1f s on %d threads, then every 16th ray on one thread in "
                      "%.1f s; CPU restatement of the reference tracer, OpenMP schedule(dynamic,1024), no per-ray stats "
                      "timers" % (passes, len(rays), dt, orc.max_threads(), dt1)}


def other_ray_kinds(args, w, device, reps=10):
    """Outside the timed headline: the other two ray populations of a frame on the same scene, each timed as whole
    device calls (sort included where the library sorts) with the traversal-kernel time from HIP events beside it:
    primary = closest_hit on this rank's jittered camera rays (what the reference's rt_bench times, driver/main.cpp:50-65),
    shadow = any_hit from every primary hit towards a point light (gi/primary-hit.cpp:43-50)."""
    import torch
    from raytracing_amd import wavefront
    t = w["tracer"]
    dev = "cuda:%d" % device
    npix = args.width * args.height
    gen, kw, cam_pos, cam_dir, cam_up = WORKLOADS[args.workload]
    cam_dir = np.asarray(cam_dir, np.float32) / np.float32(np.linalg.norm(cam_dir))
    jitter = np.stack([wavefront.uniform01(np.arange(npix), 100, 0) - 0.5, wavefront.uniform01(np.arange(npix), 101, 0) - 0.5], 1).astype(np.float32)
    d_primary = t.camera_rays_device(cam_dir, cam_pos, cam_up, 65.0, args.width, args.height, d_offsets=torch.from_numpy(jitter).to(dev))
    d_hits = torch.empty((npix, 4), dtype=torch.float32, device=dev)
    t.closest_hit_device(d_primary, d_hits)
    lo, hi = w["pos"].min(0), w["pos"].max(0)
    light = tuple(float(v) for v in (lo + hi) / 2 + (hi - lo) * np.array([0.15, 0.35, 0.2]))
    d_shadow = t.shadow_rays_device(d_hits, light)
    d_occ = torch.empty(npix, dtype=torch.uint8, device=dev)
    out = {}
    for name, call, n in (("primary", lambda: t.closest_hit_device(d_primary, d_hits), npix),
                          ("shadow", lambda: t.any_hit_device(d_shadow, d_occ), npix)):
        call()
        t.profile_begin()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        k_ms, launches = t.profile_end()
        out[name] = {"value": round(reps * n / dt / 1e6, 2), "unit": "Mrays/s", "rays": int(n), "kernel_ms": round(k_ms / max(launches, 1), 4)}
    out["shadow"]["occluded_share"] = round(float(d_occ.float().mean().item()), 4)
    out["note"] = ("same scene, %dx%d: primary = closest_hit on jittered camera rays, shadow = any_hit from every primary hit "
                   "towards a point light (rays of misses are empty); whole device calls, outside the timed headline" % (args.width, args.height))
    return out


def host_buffer_rates(t, rays, d_hits):
    """rtgi_hip_closest_hit on the same batch with rays and hits in host memory, PCIe both ways inside the call:
    once with ordinary (pageable) numpy arrays, once with page-locked buffers from rtgi_hip_host_alloc."""
    import ctypes
    from raytracing_amd import tracer
    n = len(rays)
    want = d_hits.cpu().numpy().tobytes()
    L = tracer.lib()

    def rate(rays_ptr, hits_ptr, reps=5):
        t._check(L.rtgi_hip_closest_hit(t._h, rays_ptr, n, hits_ptr))
        t0 = time.perf_counter()
        for _ in range(reps):
            t._check(L.rtgi_hip_closest_hit(t._h, rays_ptr, n, hits_ptr))
        return round(reps * n / (time.perf_counter() - t0) / 1e6, 2)

    hits = np.zeros((n, 4), np.float32)
    pageable = rate(rays.ctypes.data, hits.ctypes.data)
    assert hits.tobytes() == want, "host-buffer path disagrees with the device path"
    out = {"value": pageable, "unit": "Mrays/s",
           "note": "rtgi_hip_closest_hit on the same batch from ordinary host memory, copies included; "
                   "page_locked: the same with buffers from rtgi_hip_host_alloc"}
    p_rays, p_hits = L.rtgi_hip_host_alloc(rays.nbytes), L.rtgi_hip_host_alloc(hits.nbytes)
    if p_rays and p_hits:
        ctypes.memmove(p_rays, rays.ctypes.data, rays.nbytes)
        out["page_locked"] = rate(p_rays, p_hits)
        got = ctypes.string_at(p_hits, hits.nbytes)
        assert got == want, "host-buffer path (page-locked) disagrees with the device path"
    for ptr in (p_rays, p_hits):
        if ptr:
            L.rtgi_hip_host_free(ptr)
    return out


def gather_ceiling():
    """(16-byte lane loads per second with four loads per 64-byte record, 64-byte L2-miss records per second, source,
    lane loads per second with three loads per record or None) of this GPU, measured now by scripts/ubench/gather_mix
    if it has been built, else the figures recorded under profiles/."""
    import subprocess
    exe = os.path.join(ROOT, "scripts", "ubench", "gather_mix")
    if os.path.exists(exe):
        try:
            out = subprocess.run([exe, "4", "quick"], capture_output=True, text=True, timeout=60).stdout
            vals = dict(line.split() for line in out.splitlines() if line.startswith(("LANE_LOADS", "MISS_RECORDS")))
            three = float(vals["LANE_LOADS_PER_S_3"]) if "LANE_LOADS_PER_S_3" in vals else None
            return float(vals["LANE_LOADS_PER_S"]), float(vals["MISS_RECORDS_PER_S"]), "measured on this GPU by scripts/ubench/gather_mix", three
        except (OSError, KeyError, ValueError, subprocess.SubprocessError):
            pass
    return LANE_LOADS_PER_S_FALLBACK, MISS_RECORDS_PER_S_FALLBACK, "profiles/r02_ubench_gather_mix_3waves.txt", None


def kernel_source_hash():
    """Identifies the kernel (traversal code and node layout) the counters under profiles/ were taken with."""
    import hashlib
    h = hashlib.sha1()
    for name in ("trace_kernel.h", "bvh_host.cpp"):
        with open(os.path.join(ROOT, "raytracing_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def presort(rays, how):
    """Experiment: order rays by origin Morton code and/or direction octant."""
    o = rays[:, 0:3].astype(np.float64)
    lo, hi = o.min(0), o.max(0)
    q = np.clip(((o - lo) / np.maximum(hi - lo, 1e-9) * 1023).astype(np.uint64), 0, 1023)

    def spread(v):
        v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
        return v
    morton = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    octant = ((rays[:, 3] < 0).astype(np.uint64) | ((rays[:, 4] < 0).astype(np.uint64) << np.uint64(1))
              | ((rays[:, 5] < 0).astype(np.uint64) << np.uint64(2)))
    if how == "morton":
        key = morton
    elif how == "octant_morton":
        key = (octant << np.uint64(30)) | morton
    elif how == "morton_octant":
        key = ((morton >> np.uint64(9)) << np.uint64(3)) | octant
    else:
        key = np.random.RandomState(0).permutation(len(rays)).astype(np.uint64)
    return np.ascontiguousarray(rays[np.argsort(key, kind="stable")])


def _oracle_split(bvh):
    kw = dict(split="om", planes=0, max_tris=1)
    for c in [c.strip() for c in bvh.split(";") if c.strip()]:
        p = c.split()
        if p[0] in ("om", "sm"):
            kw["split"] = p[0]
        elif p[0] == "sah":
            kw["split"], kw["planes"] = "sah", int(p[1])
        elif p[0] == "triangles":
            kw["max_tris"] = 1 if p[1] == "single" else int(p[2])
    return kw


def whole_job(elapsed, n, where, grouped):
    """The job's time is its slowest rank's, its work the sum over the ranks (`where`: the device the process
    group's backend exchanges tensors on)."""
    import torch
    import torch.distributed as dist
    if not grouped:
        return elapsed, float(n)
    stats = torch.tensor([elapsed, float(n)], dtype=torch.float64, device=where)
    mx = stats.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = stats.clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    return float(mx[0].item()), float(sm[1].item())


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = 0 if args.one_device else local_rank
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # by torch.distributed.run, whatever the world size
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
    elif args.gpus != 1:
        raise SystemExit("--gpus %d needs to be launched through torch.distributed.run" % args.gpus)
    torch.cuda.set_device(device)
    grouped = dist.is_initialized()
    if args.frame:
        frame_mode(args, world, rank, device, grouped)
        if grouped:
            dist.barrier()
            dist.destroy_process_group()
        return

    w = build_workload(args, rank, device)
    t = w["tracer"]
    if args.presort != "none":
        w["bounce"] = presort(w["bounce"], args.presort)
    d_rays = torch.from_numpy(w["bounce"]).to("cuda:%d" % device)
    n = d_rays.shape[0]
    d_hits = torch.empty((n, 4), dtype=torch.float32, device=d_rays.device)

    alg = algorithmic_bytes(t, d_rays, args.layout)
    total_bytes, pops, tests, hit_rate = alg["bytes"], alg["pops"], alg["tests"], alg["hit_rate"]
    lane_rate, miss_rate, rate_source, lane_rate3 = gather_ceiling() if rank == 0 else (LANE_LOADS_PER_S_FALLBACK, MISS_RECORDS_PER_S_FALLBACK, "", None)

    def barrier():
        torch.cuda.synchronize(device)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        t.closest_hit_device(d_rays, d_hits)
    t.profile_begin()  # HIP events around every traversal-kernel launch, on the launch stream, inside the library
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t.closest_hit_device(d_rays, d_hits)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_total_ms, launches = t.profile_end()
    kernel_ms = kernel_total_ms / max(launches, 1)
    # the timed launches did the work: the last one's hit records are the counting launch's
    assert torch.equal(d_hits, alg["hits"]), "timed launch and counting launch disagree"

    elapsed, total_rays = whole_job(elapsed, n, "cuda:%d" % device if args.backend == "nccl" else "cpu", grouped)

    if rank == 0:
        achieved = total_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrays/sec (closest-hit, incoherent bounce)",
            "value": round(total_rays * args.steps / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "triangles": int(len(w["tri"])),
                "bvh": ("%s %s" % (args.layout, args.bvh or "om")).strip(),
                "resolution": "%dx%d" % (args.width, args.height),
                "rays_per_step_per_gpu": int(n),
                "sharding": "sample index per GPU, BVH replicated",
                "inner_pops_per_ray": round(pops, 2),
                "triangle_tests_per_ray": round(tests, 3),
                "bounce_hit_rate": round(hit_rate, 4),
                "primary_hit_rate": round(w["primary_hit_rate"], 4),
                "bvh_build_s": round(w["build_s"], 2),
            },
            "roofline": roofline(args.workload, total_bytes, n, kernel_ms, alg, lane_rate, miss_rate, rate_source, lane_rate3),
        }
        if world == 1 and not args.no_extra:
            out["extra"] = other_ray_kinds(args, w, device)
        # the drop-in form of the same call: rays and hits in host memory, PCIe both ways inside the call (never `value`)
        if not args.no_host_buffers and world == 1:
            out["host_buffer_path"] = host_buffer_rates(t, w["bounce"], d_hits)
        if not args.no_cpu and world == 1:  # the CPU leg is timed beside the one-GPU run only
            out["cpu_baseline"] = cpu_baseline(args, w, d_hits.cpu().numpy().view(np.dtype([("t", "<f4"), ("beta", "<f4"), ("gamma", "<f4"), ("ref", "<u4")])).reshape(-1))
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def frame_mode(args, world, rank, device, grouped):
    """One step = one frame of `sppx` samples per pixel, path-traced (camera ray, `bounces` diffuse bounces, a shadow
    ray at every vertex) by the C++ frame loop: every rank renders its 64x64 tiles with the whole BVH on its GPU, then
    one ncclAllReduce of the framebuffer (sums and counts), all inside the timed region."""
    import torch
    import torch.distributed as dist
    from raytracing_amd import frame, scenes
    gen, kw, cam_pos, cam_dir, cam_up = WORKLOADS[args.workload]
    pos, tri, nrm = getattr(scenes, gen)(**kw)
    cam_dir = np.asarray(cam_dir, np.float32) / np.float32(np.linalg.norm(cam_dir))
    dev = "cuda:%d" % device if args.backend == "nccl" else "cpu"
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid = torch.tensor(list(frame.HipRenderer.unique_id()), dtype=torch.uint8, device=dev)
    if grouped:
        dist.broadcast(uid, 0)
    r = frame.HipRenderer(devices=(device,), layout=args.layout, rank=rank, world=world, unique_id=bytes(uid.cpu().tolist()))
    for cmd in [c.strip() for c in args.bvh.split(";") if c.strip()]:
        r.interprete("bvh", cmd)
    t0 = time.time()
    r.build(pos, nrm, tri, np.zeros(len(tri), np.uint32), np.array([[0.7, 0.7, 0.7]], np.float32))
    build_s = time.time() - t0
    lo, hi = pos.min(0), pos.max(0)
    light = tuple(float(v) for v in (lo + hi) / 2 + (hi - lo) * np.array([0.15, 0.35, 0.2]))
    kwr = dict(cam_dir=cam_dir, cam_pos=cam_pos, cam_up=cam_up, fovy=65.0, w=args.width, h=args.height, algo="path",
               bounces=args.bounces, light_pos=light, light_col=(1.0, 1.0, 1.0), seed=1, sharding=args.sharding, want_rgb8=False)

    def barrier():
        torch.cuda.synchronize(device)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        r.render(sppx=min(args.sppx, 4), **kwr)
    rays, ms_trace, ms_reduce = 0, 0.0, 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fb, _, st = r.render(sppx=args.sppx, **kwr)
        rays += st["rays"]
        ms_trace += st["ms_trace"]
        ms_reduce += st["ms_reduce"]
    barrier()
    elapsed = time.perf_counter() - t0
    assert (fb[:, 3] == args.sppx).all(), "the combined frame does not hold every sample of every pixel"
    stats = torch.tensor([elapsed, float(rays), ms_reduce], dtype=torch.float64, device=dev)
    total_rays = float(rays)
    if grouped:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, total_rays, ms_reduce = float(mx[0].item()), float(sm[1].item()), float(mx[2].item())
    if rank == 0:
        print(json.dumps({
            "metric": "Mrays/sec (path-traced frame: closest-hit and any-hit rays, framebuffer all-reduce included)",
            "value": round(total_rays / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload.replace("_bounce", "") + "_frame", "triangles": int(len(tri)),
                       "bvh": ("%s %s" % (args.layout, args.bvh or "om")).strip(), "resolution": "%dx%d" % (args.width, args.height),
                       "sppx": args.sppx, "bounces": args.bounces, "sharding": "%s dealt over the GPUs, BVH replicated" % args.sharding,
                       "rays_per_frame": int(total_rays / args.steps), "bvh_build_s": round(build_s, 2)},
            "framebuffer_exchange": {"how": "one ncclAllReduce of w*h float4 (sums and counts) per frame, called from C++",
                                     "bytes": args.width * args.height * 16, "ms_per_frame_max_over_ranks": round(ms_reduce / args.steps, 3),
                                     "used_rccl": int(st["used_rccl"])},
            "mean_radiance": [round(float(v), 6) for v in fb[:, 0:3].mean(0)],
        }), flush=True)
    r.close()


def roofline(workload, total_bytes, n, kernel_ms, alg, lane_rate, miss_rate, rate_source, lane_rate3=None):
    """What the dominant kernel runs against, and how close it is.

    `achieved` is the reference's algorithmic bytes per launch over the kernel time, measured here with HIP events on the
    launch stream. Against the HBM peak that figure exceeds 1 (`algorithmic_bytes_over_hbm_peak`), because the tree lives
    in the caches; so three kinds of figures stand beside it, each recomputable from the files under profiles/:
      frac_hbm_traffic  bytes the counters saw beyond L2 per launch / kernel time / 8 TB/s
      frac_l2           64-byte requests from the vector caches to L2 per launch x 64 / kernel time / 34.5 TB/s
      pipeline_busy     busy and stall shares of the units of the vector memory pipeline (TA: texture addresser, TCP:
                        vector cache, TD: data return) and of the SIMDs, as the hardware counters report them
      frac              the largest of the times each resource needs for this launch's work alone / kernel time, with
                        `bound` naming that resource and `peak` = the algorithmic bytes over that time. Resources:
        lane loads   16-byte loads of lanes going to different lines (counted by the kernel itself) / measured rate; the
                     rate is measured with four loads per 64-byte record (visits that fetch their links) and with three
                     (all other visits, triangles), and each kind of load is priced at its own
        valu         vector instructions x 4 cycles on 1024 SIMDs
        beyond L2    L2 misses / measured rate of random 64-byte fetches that miss L2
        hbm          bytes beyond L2 / 8 TB/s
        chain        the longest ray's inner-node visits x the latency of one dependent fetch
    The counters under profiles/ are used only if they were taken with this kernel source (hash) and workload."""
    t_kernel = kernel_ms * 1e-3
    loads4 = min(4 * alg["link_fetches"], alg["lane_loads"]) if lane_rate3 else alg["lane_loads"]
    times = {"lane_loads": loads4 / lane_rate + (alg["lane_loads"] - loads4) / (lane_rate3 or lane_rate), "chain": alg["max_pops"] * 355e-9}
    pmc = _measured_counters(workload)
    if pmc:
        times["valu"] = pmc["valu_instructions"] * VALU_CYCLES_PER_INSTRUCTION / SIMDS / (pmc["clock_ghz"] * 1e9)
        times["beyond_l2"] = pmc["l2_misses"] / miss_rate
        times["hbm"] = pmc["bytes_beyond_l2"] / (HBM_PEAK_GBS * 1e9)
    bound = max(times, key=times.get)
    names = {"lane_loads": "vector memory pipeline: 16-byte loads of lanes that go to different cache lines",
             "valu": "vector instruction issue", "beyond_l2": "random 64-byte fetches that miss L2", "hbm": "hbm",
             "chain": "latency of the longest ray's chain of dependent fetches"}
    achieved = total_bytes / t_kernel / 1e9
    frac = times[bound] / t_kernel
    out = {
        "bound": names[bound],
        "achieved": round(achieved, 1),
        "peak": round(achieved / frac, 1),
        "unit": "GB/s",
        "frac": round(frac, 4),
        "traffic": pmc["bytes_beyond_l2"] if pmc else None,
        "frac_hbm_traffic": round(pmc["bytes_beyond_l2"] / t_kernel / (HBM_PEAK_GBS * 1e9), 4) if pmc else None,
        "frac_l2": round(pmc["tcp_to_l2_read_requests"] * 64 / t_kernel / (L2_PEAK_GBS * 1e9), 4) if pmc else None,
        "pipeline_busy": pmc.get("busy") if pmc else None,
        "kernel": "rtgi_dev::trace_kernel<closest_hit> (ray sort and counter reset excluded)",
        "kernel_ms": round(kernel_ms, 4),
        "algorithmic_bytes_per_launch": int(total_bytes),
        "algorithmic_bytes_per_ray": round(total_bytes / n, 1),
        "algorithmic_bytes_over_hbm_peak": round(achieved / HBM_PEAK_GBS, 4),
        "floor_ms_by_resource": {k: round(v * 1e3, 4) for k, v in sorted(times.items())},
        "lane_loads_per_launch": alg["lane_loads"],
        "lane_load_rate_per_s": lane_rate,
        "lane_load_rate_three_per_record_per_s": lane_rate3,
        "lane_load_rate_source": rate_source,
        "inner_visits_served_from_lds": round(alg["lds_share"], 4),
        "inner_visits_that_fetch_links": round(alg["link_fetch_share"], 4),
        "counters": ("profiles/traffic.json" if pmc else "none for kernel source %s: traffic, fractions, busy shares, valu, beyond_l2 and hbm not priced" % kernel_source_hash()),
    }
    if pmc:
        out["waves"] = {k: pmc[k] for k in ("wave_cycles_executing", "wave_cycles_waiting_for_issue", "wave_cycles_waiting_for_anything",
                                            "l2_request_latency_cycles") if pmc.get(k) is not None}
    return out


def _measured_counters(workload):
    """Per-launch hardware counters of the traversal kernel from the rocprofv3 PMC passes kept under profiles/
    (scripts/pmc_bench.sh + scripts/pmc_to_traffic.py), or None if they belong to another kernel source."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(workload)
    except (OSError, ValueError):
        return None
    if not isinstance(entry, dict) or entry.get("kernel_source") != kernel_source_hash():
        return None
    return entry


if __name__ == "__main__":
    main()
