"""Experiment: how much of the launch's tail is the length predictor's fault?

The library orders a batch by (direction octant | length class | Morton code of the origin), the class being an
upper bound of the ray's length (its span inside the scene's bounds). This script orders the benchmark batch on the
host instead, once with the same key and once with the class taken from the number of nodes every ray really
visits (known from a counting launch: a predictor nobody can have), and traces both in the order given
("hip reorder off"). The difference between the two is what a perfect predictor could win.

Run on the GPU box: python scripts/experiments/known_length_order.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def spread3(v):
    v = (v | (v << 8)) & 0x0000F00F
    v = (v | (v << 4)) & 0x000C30C3
    v = (v | (v << 2)) & 0x00249249
    return v


def keys(rays, lo, hi, cls, class_bits):
    o = rays[:, 0:3]
    cell = np.clip(((o - lo) / (hi - lo) * 64.0), 0, 63).astype(np.uint32)
    morton = spread3(cell[:, 0]) | (spread3(cell[:, 1]) << 1) | (spread3(cell[:, 2]) << 2)
    octant = (rays[:, 3] < 0).astype(np.uint32) | ((rays[:, 4] < 0).astype(np.uint32) << 1) | ((rays[:, 5] < 0).astype(np.uint32) << 2)
    return (octant.astype(np.uint64) << np.uint64(40)) | (cls.astype(np.uint64) << np.uint64(20)) | morton.astype(np.uint64)


def span_class(rays, lo, hi, classes):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / rays[:, 3:6]
        ta, tb = (lo - rays[:, 0:3]) * inv, (hi - rays[:, 0:3]) * inv
    t_in = np.maximum(rays[:, 6], np.minimum(ta, tb).max(1))
    t_out = np.minimum(rays[:, 7], np.maximum(ta, tb).min(1))
    rel = np.nan_to_num((t_out - t_in) / np.linalg.norm(hi - lo) * classes, nan=0.0)
    return (classes - 1) - np.clip(rel, 0, classes - 1).astype(np.uint32)


def main():
    import argparse
    import torch
    import bench
    args = argparse.Namespace(workload="soup1m_bounce", layout="flat", bvh="", width=1920, height=1080)
    w = bench.build_workload(args, 0, 0)
    t, rays = w["tracer"], w["bounce"]
    n = len(rays)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    d_counts = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    t.closest_hit_device(d_rays, d_hits, d_counts)
    torch.cuda.synchronize()
    pops = d_counts[:, 0].cpu().numpy().astype(np.int64)
    want = d_hits.clone()
    lo, hi = w["pos"].min(0), w["pos"].max(0)

    def timed(name, order, reorder):
        t.interprete("hip", "reorder " + reorder)
        r = torch.from_numpy(np.ascontiguousarray(rays[order])).cuda() if order is not None else d_rays
        for _ in range(3):
            t.closest_hit_device(r, d_hits)
        t.profile_begin()
        for _ in range(10):
            t.closest_hit_device(r, d_hits)
        torch.cuda.synchronize()
        ms, launches = t.profile_end()
        if order is not None:  # same hit records, in the order traced
            assert torch.equal(d_hits, want[torch.from_numpy(order).cuda()])
        print("%-58s kernel %.3f ms" % (name, ms / launches), flush=True)

    timed("library order (sorted on the GPU)", None, "on")
    for classes in (32, 256):
        span = span_class(rays, lo, hi, classes)
        timed("host order, %3d classes from the span (the library's key)" % classes, np.argsort(keys(rays, lo, hi, span, 0), kind="stable"), "off")
        # classes of equal width in visited nodes, longest first
        known = (classes - 1) - np.minimum(pops * classes // (pops.max() + 1), classes - 1).astype(np.uint32)
        timed("host order, %3d classes from the visits counted" % classes, np.argsort(keys(rays, lo, hi, known, 0), kind="stable"), "off")
    print("correlation of span and visits: %.3f" % np.corrcoef(span_class(rays, lo, hi, 1 << 16).astype(np.float64), pops)[0, 1])


if __name__ == "__main__":
    main()
