#!/usr/bin/env python3
"""Randomised parity campaign: random scenes, tree settings and ray mixes, GPU against the CPU oracle, bit for bit
(hits, any-hit bytes, per-ray counts; the counting and the plain instantiation of the kernel, and for large batches the
host-buffer entry point on page-locked memory). Runs until the time budget is used; prints one line per configuration and
stops at the first difference.

    python scripts/fuzz_parity.py [--seconds 120] [--seed 0]
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
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    from oracle import oracle as orc
    from raytracing_amd import scenes, tracer
    from tests.test_gpu_parity import _mixed_rays
    from tests.util import assert_hits_equal
    rs = np.random.RandomState(args.seed)
    t_end = time.time() + args.seconds
    done = 0
    while time.time() < t_end:
        kind = rs.choice(["soup", "cornell", "sponza", "terrain", "skewed", "stadium"])
        size = int(rs.choice([300, 3000, 30000, 150000]))
        seed = int(rs.randint(0, 1 << 30))
        if kind == "soup":
            pos, tri, _ = scenes.soup(size, seed=seed, size=float(rs.choice([0.02, 0.2, 2.0])))
        elif kind == "cornell":
            pos, tri, _ = scenes.cornell(int(rs.randint(1, 12)))
        elif kind == "sponza":
            pos, tri, _ = scenes.sponza_class(max(size, 3000), seed=seed)
        elif kind == "terrain":
            pos, tri, _ = scenes.terrain_objects(max(size, 3000), seed=seed)
        elif kind == "stadium":  # a chain of sheets: deep trees under sm, stacks beyond the part kept in LDS
            sheets = int(rs.choice([26, 34, 44]))
            pos, tri, _ = scenes.stadium_sheets(max(size, 3000), sheets=sheets, seed=seed)
        else:
            pos, tri, _ = scenes.skewed(size, seed=seed)
        layout = str(rs.choice(["flat", "indexed"]))
        esc = bool(layout == "indexed" and rs.rand() < 0.25)
        split = str(rs.choice(["om", "sm", "sah"]))
        planes = int(rs.choice([1, 4, 16, 63]))
        max_tris = int(rs.choice([1, 1, 2, 5, 16]))
        n = int(rs.choice([1000, 70000, 200000, 560000]))
        reorder = str(rs.choice(["auto", "on", "off"]))
        t = tracer.HipBvhTracer(layout=layout, esc=esc)
        if split == "sah":
            t.interprete("bvh", "sah %d" % planes)
        else:
            t.interprete("bvh", split)
        t.interprete("bvh", "triangles single" if max_tris == 1 else "triangles multiple %d" % max_tris)
        t.interprete("hip", "reorder " + reorder)
        label = "%s %d tris, %s esc=%d %s planes=%d leaf=%d, %d rays, reorder %s" % (
            kind, len(tri), layout, esc, split, planes, max_tris, n, reorder)
        try:
            t.build(pos, tri)
        except tracer.RtgiHipError as e:  # early split clipping refuses degenerate triangles
            print("skipped (%s): %s" % (e, label), flush=True)
            continue
        o = orc.Bbvh(pos, tri, layout=layout, esc=esc, split=split, planes=planes if split == "sah" else 0, max_tris=max_tris)
        assert t.nodes().tobytes() == o.nodes().tobytes(), "tree differs: " + label
        rays = _mixed_rays(pos, n, seed=seed % 1000)
        if kind == "stadium":  # half of the rays along the chain
            rays[::2] = scenes.stadium_rays((n + 1) // 2, seed=seed % 1000, sheets=sheets)
        d_rays = torch.from_numpy(rays).cuda()
        d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
        t.closest_hit_device(d_rays, d_hits, d_cnt)
        want, want_counts = o.closest_hit(rays, counts=True)
        assert_hits_equal(d_hits.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1), want, label)
        assert (d_cnt.cpu().numpy().astype(np.uint32) == want_counts[:, 0:2]).all(), "counts differ: " + label
        d_cnt.zero_()
        t.any_hit_device(d_rays, d_occ, d_cnt)
        want_any, want_any_counts = o.any_hit(rays, counts=True)
        assert (d_occ.cpu().numpy() == want_any).all(), "any_hit differs: " + label
        assert (d_cnt.cpu().numpy().astype(np.uint32) == want_any_counts[:, 0:2]).all(), "any_hit counts differ: " + label
        # the instantiation without counters (the one that is timed)
        d_hits.fill_(-1.0)
        d_occ.fill_(7)
        t.closest_hit_device(d_rays, d_hits)
        t.any_hit_device(d_rays, d_occ)
        assert_hits_equal(d_hits.cpu().numpy().view(tracer.HIT_DTYPE).reshape(-1), want, "plain kernel: " + label)
        assert (d_occ.cpu().numpy() == want_any).all(), "plain any_hit differs: " + label
        if n >= 500000:  # host buffers in page-locked memory: one launch that takes up the batch as it arrives
            import ctypes
            L = tracer.lib()
            p_rays, p_hits = L.rtgi_hip_host_alloc(rays.nbytes), L.rtgi_hip_host_alloc(n * 16)
            assert p_rays and p_hits
            ctypes.memmove(p_rays, rays.ctypes.data, rays.nbytes)
            ctypes.memset(p_hits, 0xAB, n * 16)
            t._check(L.rtgi_hip_closest_hit(t._h, p_rays, n, p_hits))
            got = np.frombuffer(ctypes.string_at(p_hits, n * 16), dtype=tracer.HIT_DTYPE)
            L.rtgi_hip_host_free(p_rays), L.rtgi_hip_host_free(p_hits)
            assert_hits_equal(got, want, "page-locked host batch: " + label)
            label += ", host path %s" % t.host_path_stats()
        t.close()
        done += 1
        print("ok  depth %2d  %s" % (o.depth, label), flush=True)
    print("%d configurations, no difference" % done)


if __name__ == "__main__":
    main()
