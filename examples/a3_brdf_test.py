#!/usr/bin/env python3
"""The reference's scripts/a3-brdf-test frame ("Assignment 3/rtgi-2021-a03-ref/") through the GPU wavefront
pipeline: camera rays -> closest_hit -> shadow rays -> any_hit -> local illumination -> framebuffer -> PNG.

    python examples/a3_brdf_test.py [--out a3-brdf-test.png] [--sppx 16] [--width 1280 --height 720]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/a3_brdf_test.py ...

With N ranks every GPU holds the whole scene and renders every N-th sample; the framebuffers (running mean and
sample count per pixel) are combined by one all-reduce of sums and counts, and rank 0 writes the file.

Everything between the OBJ file and the PNG file runs on the GPU; rays and hits never leave it. All materials use
the lambertian BRDF (the script's default-brdf) without textures.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--obj", default=os.path.join(ROOT, "tests", "golden", "render-data", "brdf-test-noal.obj"))
    ap.add_argument("--out", default="a3-brdf-test.png")
    ap.add_argument("--sppx", type=int, default=16)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from raytracing_amd import distributed, png, scenes, tracer
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    device = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
    s = scenes.load_obj_materials(args.obj)
    kd, ks = s["kd"], s["ks"]
    luma = 0.212671 * kd[:, 0] + 0.715160 * kd[:, 1] + 0.072169 * kd[:, 2]
    albedo = np.power(np.where((luma > 1e-4)[:, None], kd, ks).astype(np.float32), np.float32(2.2))  # libgi/scene.cpp:123-125
    # at / look / camup, pointlight, resolution and sppx of the script
    cam_pos, cam_dir, cam_up = (-2.4008, 1.7537, -0.050779), (0.85235, -0.52249, -0.022323), (0.52233, 0.85264, -0.013184)
    light_pos, light_col = (2.0, 1.5, 0.0), (4.0, 2.0, 1.0)
    w, h = args.width, args.height
    t = tracer.HipBvhTracer(layout="flat", device=device)  # raytracer bbvh
    t.build(s["pos"], s["tri"])             # commit
    t.set_shading(s["normals"], s["tri"], s["material_id"], albedo)
    fb = distributed.Framebuffer(w, h, device="cuda:%d" % device)
    d_fb = fb.color.view(w * h, 4)
    d_hits = torch.empty((w * h, 4), dtype=torch.float32, device=d_fb.device)
    d_occ = torch.empty(w * h, dtype=torch.uint8, device=d_fb.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for sample in distributed.shard_samples(args.sppx, world, rank):  # algo local; run
        d_rays = t.camera_rays_device(cam_dir, cam_pos, cam_up, 65.0, w, h, d_offsets=t.pixel_offsets_device(w * h, sample))
        t.closest_hit_device(d_rays, d_hits)
        t.any_hit_device(t.shadow_rays_device(d_hits, light_pos), d_occ)
        t.framebuffer_add_device(d_fb, t.shade_local_device(d_hits, d_occ, light_pos, light_col))
    if world > 1:
        if args.backend == "gloo":  # gloo reduces host tensors
            host = distributed.Framebuffer(w, h)
            host.color.copy_(fb.color)
            fb.color.copy_(host.reduce().color)
        else:
            fb.reduce()
    rgb = t.framebuffer_rgb8_device(d_fb, w, h).cpu().numpy()
    dt = time.perf_counter() - t0
    if rank == 0:
        png.write_rgb8(args.out, rgb)
        print("%dx%d, %d sppx on %d GPU(s): %.1f ms, %.0f Mrays/s (camera + shadow rays); wrote %s" %
              (w, h, args.sppx, world, dt * 1e3, 2 * w * h * args.sppx / dt / 1e6, args.out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
