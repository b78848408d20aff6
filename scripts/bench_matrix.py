#!/usr/bin/env python3
"""Throughput of every ray population on every benchmark scene, GPU next to the CPU oracle.

    python scripts/bench_matrix.py [--cases a,b,...] [--out gpurun_out/matrix.jsonl] [--cpu-seconds S]

For each case: build the tree, then on the GPU generate and trace primary rays (closest_hit), shadow rays
towards a point light (any_hit) and one diffuse bounce (closest_hit). Each population is timed over whole
calls of the device entry point (key generation + sort + traversal) and, separately, over the traversal
kernel alone; the kernel counts inner-node pops and triangle tests, from which the bytes the reference's data
structures would move follow (32 B ray + 16 B hit | 1 B flag + 56 B per pop + 52 B per test, + 4 B per test
through index[]). The CPU legs run the oracle on a bounded prefix sample with all threads and with one.
One JSON line per (case, population); nothing here is the headline number of bench.py.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    # name: scene generator, kwargs, layout, bvh commands, camera (pos, dir, up), resolution, light
    "cornell": ("cornell", dict(n=1), "flat", "", ((0, 0, 3.4), (0, 0, -1), (0, 1, 0)), (1280, 720), (0.0, 0.9, 0.0)),
    "sponza260k": ("sponza_class", dict(target_tris=262144, seed=0), "flat", "",
                   ((-516, 300, 0), (1, 0, 0), (0, 1, 0)), (1280, 720), (0.0, 900.0, 0.0)),
    "soup1m": ("soup", dict(n_tris=1 << 20, seed=1), "flat", "", ((0, 0, 0), (0, 0, -1), (0, 1, 0)), (1920, 1080),
               (3.0, 8.0, 5.0)),
    "terrain1m": ("terrain_objects", dict(target_tris=1 << 20, seed=3), "flat", "",
                  ((0, 6, 14), (0, -0.35, -0.937), (0, 1, 0)), (1920, 1080), (4.0, 25.0, 3.0)),
    "soup1m_sah": ("soup", dict(n_tris=1 << 20, seed=1), "indexed", "sah 16; triangles multiple 4",
                   ((0, 0, 0), (0, 0, -1), (0, 1, 0)), (1920, 1080), (3.0, 8.0, 5.0)),
    "skewed4m_sm": ("skewed", dict(n_tris=1 << 22, seed=5), "indexed", "sm", ((0, 0.5, 12), (0, 0, -1), (0, 1, 0)),
                    (1920, 1080), (2.0, 9.0, 4.0)),
    # 40 sheets of doubling size under spatial-median splits: a chain of 40 levels, deeper than the part of the stack
    # that lives in LDS; the "grazing" population runs along the chain and keeps a node pending per sheet ahead
    "stadium4m_sm": ("stadium_sheets", dict(n_tris=1 << 22, sheets=40), "indexed", "sm", ((2.0, 0.0, 0.0), (-1, 0, 0), (0, 1, 0)),
                     (1920, 1080), (1.0, 0.3, 0.2)),
}
# further populations of a case: name -> (generator in scenes, kwargs)
EXTRA_RAYS = {"stadium4m_sm": {"grazing": ("stadium_rays", dict(n=1 << 20, seed=3, sheets=40))}}


def time_calls(fn, sync, seconds=0.25, min_calls=3):
    fn()
    sync()
    calls, t0 = 0, time.perf_counter()
    while True:
        fn()
        calls += 1
        if calls >= min_calls:
            sync()
            dt = time.perf_counter() - t0
            if dt >= seconds:
                return dt / calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "matrix.jsonl"))
    ap.add_argument("--cpu-seconds", type=float, default=3.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import torch
    from raytracing_amd import scenes, tracer
    from bench import _oracle_split
    dev = "cuda:0"
    torch.cuda.set_device(0)

    def sync():
        torch.cuda.synchronize(0)

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    out = open(args.out, "a")
    for name in args.cases.split(","):
        gen, kw, layout, bvh, cam, (w, h), light = CASES[name]
        pos, tri, _ = getattr(scenes, gen)(**kw)
        t = tracer.HipBvhTracer(layout=layout, device=0)
        for cmd in [c.strip() for c in bvh.split(";") if c.strip()]:
            t.interprete("bvh", cmd)
        t0 = time.time()
        t.build(pos, tri)
        build_s = time.time() - t0
        print("[%s] %d triangles, %s %s, depth %d, build %.1f s, incoherent ray cost %.1f pops" %
              (name, len(tri), layout, bvh or "om", t.depth, build_s, t.incoherent_ray_cost), flush=True)
        cam_pos, cam_dir, cam_up = cam
        cam_dir = np.asarray(cam_dir, np.float32) / np.float32(np.linalg.norm(cam_dir))
        d_primary = t.camera_rays_device(cam_dir, cam_pos, cam_up, 65.0, w, h)
        d_hits = torch.empty((w * h, 4), dtype=torch.float32, device=dev)
        t.closest_hit_device(d_primary, d_hits)
        d_shadow = t.shadow_rays_device(d_hits, light)
        d_bounce, _ = t.bounce_rays_device(d_primary, d_hits, seed=0, bounce=1)
        d_bounce = d_bounce.contiguous()
        sync()
        oracle = None
        if not args.no_cpu:
            from oracle import oracle as orc
            t0 = time.time()
            oracle = orc.Bbvh(pos, tri, layout=layout, **_oracle_split(bvh))
            print("[%s] oracle build %.1f s" % (name, time.time() - t0), flush=True)
        populations = [("primary", d_primary, False), ("shadow", d_shadow, True), ("bounce", d_bounce, False)]
        for pop, (gen_rays, kw_rays) in EXTRA_RAYS.get(name, {}).items():
            populations.append((pop, torch.from_numpy(getattr(scenes, gen_rays)(**kw_rays)).to(dev), False))
        stack = t.profile_details()
        for pop, d_rays, any_hit in populations:
            n = d_rays.shape[0]
            if n == 0:
                continue
            d_cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
            t.profile_begin()
            if any_hit:
                d_out = torch.empty((n,), dtype=torch.uint8, device=dev)
                call = lambda: t.any_hit_device(d_rays, d_out)
                t.any_hit_device(d_rays, d_out, d_cnt)
            else:
                d_out = torch.empty((n, 4), dtype=torch.float32, device=dev)
                call = lambda: t.closest_hit_device(d_rays, d_out)
                t.closest_hit_device(d_rays, d_out, d_cnt)
            sync()
            spill_pushes = t.profile_details()["spill_pushes"]
            t.profile_end()
            s = d_cnt.sum(0, dtype=torch.int64).cpu().numpy()
            per_tri = 52 + (4 if layout == "indexed" else 0)
            total = (32 + (1 if any_hit else 16)) * n + 56 * int(s[0]) + per_tri * int(s[1])
            if any_hit:
                positive = float((d_out != 0).float().mean().item())
            else:
                positive = float((d_out[:, 0] != tracer.FLT_MAX).float().mean().item())
            for _ in range(2):
                call()
            t.profile_begin()
            per_call = time_calls(call, sync)
            kernel_total_ms, launches = t.profile_end()
            kernel_ms = kernel_total_ms / max(launches, 1)
            rec = {"case": name, "population": pop, "entry": "any_hit" if any_hit else "closest_hit", "rays": int(n),
                   "triangles": int(len(tri)), "bvh": ("%s %s" % (layout, bvh or "om")), "depth": int(t.depth),
                   "gpu_mrays_s": round(n / per_call / 1e6, 1), "ms_per_call": round(per_call * 1e3, 4),
                   "kernel_ms": round(kernel_ms, 4), "inner_pops_per_ray": round(float(s[0]) / n, 2),
                   "triangle_tests_per_ray": round(float(s[1]) / n, 3), "hit_or_occluded_fraction": round(positive, 4),
                   "algorithmic_bytes_per_ray": round(total / n, 1),
                   "stack_levels_lds": stack["lds_levels"], "stack_levels_spill": stack["spill_levels"],
                   "spill_pushes_per_ray": round(spill_pushes / n, 3),
                   "roofline_frac_of_8TBs": round(total / (kernel_ms * 1e-3) / 8e12, 4)}
            if oracle is not None:
                rays = d_rays.cpu().numpy()
                fn = oracle.any_hit if any_hit else oracle.closest_hit
                # all threads: doubling prefix until the sample takes long enough
                m = min(n, 1 << 16)
                fn(rays[:m])
                while True:
                    t0 = time.perf_counter()
                    fn(rays[:m])
                    dt = time.perf_counter() - t0
                    if dt >= args.cpu_seconds / 2 or m == n:
                        break
                    m = min(n, m * 4 if dt < args.cpu_seconds / 8 else m * 2)
                rec["cpu_mrays_s"] = round(m / dt / 1e6, 4)
                rec["cpu_threads"] = orc.max_threads()
                rec["cpu_sample"] = "first %d rays, %.2f s" % (m, dt)
                # one thread, on a strided sample an all-thread pass would take ~1/8 s for
                k = max(1024, int(m / dt / 8 / orc.max_threads()))
                sub = np.ascontiguousarray(rays[:: max(1, n // k)][:k])
                t0 = time.perf_counter()
                fn(sub, threads=1)
                dt1 = time.perf_counter() - t0
                rec["cpu_1thread_mrays_s"] = round(len(sub) / dt1 / 1e6, 4)
                rec["gpu_over_cpu"] = round(rec["gpu_mrays_s"] / rec["cpu_mrays_s"], 1)
            line = json.dumps(rec)
            print(line, flush=True)
            out.write(line + "\n")
            out.flush()
        t.close()
        del oracle
    out.close()


if __name__ == "__main__":
    main()
