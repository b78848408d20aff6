#!/usr/bin/env python3
"""Experiment: how much of the ray sort's effect survives if rays are ordered only inside windows of W consecutive rays of
the caller's order (what a one-kernel, workgroup-local sort could do), instead of over the whole batch?

The benchmark batch is ordered on the host with the library's key (octant | length class | Morton code of the origin),
once over the whole batch and once per window, and traced in the order given (`hip reorder off`); the kernel time of each
order stands beside the library's own sorted launch.   python scripts/experiments/windowed_order.py [bench.py flags]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def keys_of(rays, lo, hi, class_bits=5):
    o, d = rays[:, 0:3].astype(np.float32), rays[:, 3:6].astype(np.float32)
    inv_extent = (1.0 / np.maximum(hi - lo, 1e-30)).astype(np.float32)
    cell = np.clip(np.nan_to_num((o - lo) * inv_extent * 64.0), 0, 63).astype(np.uint32)

    def spread3(v):
        v = (v | (v << 8)) & 0x0000F00F
        v = (v | (v << 4)) & 0x000C30C3
        v = (v | (v << 2)) & 0x00249249
        return v
    morton = spread3(cell[:, 0]) | (spread3(cell[:, 1]) << 1) | (spread3(cell[:, 2]) << 2)
    octant = (d[:, 0] < 0).astype(np.uint32) | ((d[:, 1] < 0).astype(np.uint32) << 1) | ((d[:, 2] < 0).astype(np.uint32) << 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (1.0 / d).astype(np.float32)
        ta, tb = (lo - o) * inv, (hi - o) * inv
    t_in = np.maximum(rays[:, 6], np.minimum(ta, tb).max(1))
    t_out = np.minimum(rays[:, 7], np.maximum(ta, tb).min(1))
    top = float((1 << class_bits) - 1)
    rel = np.nan_to_num((t_out - t_in) / np.float32(np.linalg.norm(hi - lo)) * (top + 1.0))
    cls = (top - np.clip(rel, 0, top)).astype(np.uint32)
    return (octant << 21) | (cls << (21 - class_bits)) | (morton >> (class_bits - 3))


def main():
    import torch
    args = bench.parse_args()
    w = bench.build_workload(args, 0, 0)
    t, rays = w["tracer"], w["bounce"]
    n = len(rays)
    lo, hi = w["pos"].min(0).astype(np.float32), w["pos"].max(0).astype(np.float32)
    key = keys_of(rays, lo, hi)

    def timed(order, reorder, label):
        r = np.ascontiguousarray(rays[order]) if order is not None else rays
        d_rays = torch.from_numpy(r).cuda()
        d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        assert t.interprete("hip", "reorder " + reorder)
        for _ in range(3):
            t.closest_hit_device(d_rays, d_hits)
        torch.cuda.synchronize()
        t.profile_begin()
        for _ in range(10):
            t.closest_hit_device(d_rays, d_hits)
        torch.cuda.synchronize()
        ms, launches = t.profile_end()
        print("%-58s kernel %.4f ms" % (label, ms / launches), flush=True)

    timed(None, "on", "library sort (keys, radix sort, octant segments)")
    timed(None, "off", "caller's order, no sort")
    timed(np.argsort(key, kind="stable"), "off", "whole batch ordered by the key on the host")
    for parts in ((0.12, 0.4, 1.0), (0.5, 1.0), (0.2, 0.4, 0.6, 0.8, 1.0)):  # the chunks of a one-launch host batch, each ordered on its own
        ends = [int(n * f) & ~63 if f < 1.0 else n for f in parts]
        begins = [0] + ends[:-1]
        order = np.concatenate([b + np.argsort(key[b:e], kind="stable") for b, e in zip(begins, ends)])
        timed(order, "off", "parts ending at %s of the batch, each ordered by the key" % (parts,))
    for wsize in (131072, 524288):
        order = np.concatenate([s + np.argsort(key[s:s + wsize], kind="stable") for s in range(0, n, wsize)])
        timed(order, "off", "windows of %6d rays ordered by the key" % wsize)
        coarse = key >> 16  # octant | length class
        order = np.concatenate([s + np.argsort(coarse[s:s + wsize], kind="stable") for s in range(0, n, wsize)])
        timed(order, "off", "windows of %6d rays ordered by octant and length class" % wsize)
        order = np.concatenate([s + np.argsort(key[s:s + wsize] >> 21, kind="stable") for s in range(0, n, wsize)])
        timed(order, "off", "windows of %6d rays ordered by octant" % wsize)


if __name__ == "__main__":
    main()
