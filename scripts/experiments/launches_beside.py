#!/usr/bin/env python3
"""Experiment: what does a traversal launch lose to other kernels that start and end on another stream while it runs?
The benchmark batch is traced (device-resident, sorted) alone, then beside a chain of k empty one-thread kernels, beside the
same number of small streaming kernels, and beside radix sorts of 0.5 M unrelated keys, all on a second stream.
   python scripts/experiments/launches_beside.py [bench.py flags]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    args = bench.parse_args()
    w = bench.build_workload(args, 0, 0)
    t, rays = w["tracer"], w["bounce"]
    n = len(rays)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    one = torch.zeros(1, device="cuda")
    small = torch.zeros(1 << 20, device="cuda")
    keys = torch.randint(0, 1 << 24, (1 << 19,), device="cuda", dtype=torch.int32)

    marks = []

    def beside(kind, k):
        with torch.cuda.stream(side):
            for _ in range(k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if kind == "empty":
                    one.add_(1.0)
                elif kind == "stream":
                    small.add_(1.0)
                elif kind == "sort":
                    torch.sort(keys)
                e1.record()
                marks.append((e0, e1))

    for kind, k in (("none", 0), ("empty", 8), ("empty", 24), ("empty", 100), ("stream", 24), ("stream", 100), ("sort", 1), ("sort", 3), ("sort", 6), ("sort-alone", 6)):
        for _ in range(2):
            t.closest_hit_device(d_rays, d_hits)
        torch.cuda.synchronize()
        t.profile_begin()
        reps = 6
        t0 = time.perf_counter()
        for _ in range(reps):
            if kind != "sort-alone":
                t.closest_hit_device(d_rays, d_hits)   # keys + sort + launch on torch's current stream
            beside("sort" if kind == "sort-alone" else kind, k)  # enqueued right behind it, on the other stream
            torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        ms, launches = t.profile_end()
        each = [a.elapsed_time(b) for a, b in marks]
        del marks[:]
        print("beside %-6s x %3d: traversal kernel %.4f ms, call + side work %.4f ms, one side kernel (chain) %.4f ms on average, %.4f at most" %
              (kind, k, ms / max(launches, 1), wall * 1e3, sum(each) / max(len(each), 1), max(each + [0.0])), flush=True)


if __name__ == "__main__":
    main()
