#!/usr/bin/env python3
"""What sharing a launch is worth: the shadow rays (any_hit) and the bounce rays (closest_hit) of one sample of the
2^20-triangle soup, traced by two calls and by one rtgi_hip_trace_batches_device call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from raytracing_amd import scenes, tracer
    pos, tri, _ = scenes.soup(1 << 20, seed=1)
    t = tracer.HipBvhTracer(layout="flat")
    t.build(pos, tri)
    w, h = 1920, 1080
    d_primary = t.camera_rays_device((0, 0, -1), (0, 0, 0), (0, 1, 0), 65.0, w, h, d_offsets=t.pixel_offsets_device(w * h, 0))
    d_hits = torch.empty((w * h, 4), dtype=torch.float32, device="cuda")
    t.closest_hit_device(d_primary, d_hits)
    d_shadow = t.shadow_rays_device(d_hits, (3.0, 8.0, 5.0))
    d_bounce = t.bounce_rays_device(d_primary, d_hits, seed=0, bounce=1)[0].contiguous()
    d_occ = torch.empty(w * h, dtype=torch.uint8, device="cuda")
    d_bhits = torch.empty((d_bounce.shape[0], 4), dtype=torch.float32, device="cuda")

    def separate():
        t.any_hit_device(d_shadow, d_occ)
        t.closest_hit_device(d_bounce, d_bhits)

    def together():
        t.trace_batches_device([(d_shadow, d_occ, "any"), (d_bounce, d_bhits, "closest")])
    separate()
    torch.cuda.synchronize()
    want = (d_occ.clone(), d_bhits.clone())
    together()
    torch.cuda.synchronize()
    assert torch.equal(want[0], d_occ) and torch.equal(want[1], d_bhits)
    n = w * h + d_bounce.shape[0]
    for rep in range(2):
        for name, fn in (("two calls", separate), ("one launch", together)):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print("%-10s %.3f ms  %.0f Mrays/s" % (name, dt * 1e3, n / dt / 1e6), flush=True)


if __name__ == "__main__":
    main()
