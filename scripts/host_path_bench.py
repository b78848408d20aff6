#!/usr/bin/env python3
"""Rate of the host-buffer entry point (rtgi_hip_closest_hit: rays and hits in host memory, PCIe both ways inside the
call) on bench.py's batch, from ordinary and from page-locked memory. Takes bench.py's workload flags."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    args = bench.parse_args()
    w = bench.build_workload(args, 0, 0)
    tracer, rays = w["tracer"], w["bounce"]
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((len(rays), 4), dtype=torch.float32, device="cuda")
    tracer.closest_hit_device(d_rays, d_hits)  # the reference result host_buffer_rates checks against
    torch.cuda.synchronize()
    print(bench.host_buffer_rates(tracer, rays, d_hits))


if __name__ == "__main__":
    main()
