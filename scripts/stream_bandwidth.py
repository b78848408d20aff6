#!/usr/bin/env python3
"""Achievable HBM bandwidth of the box, for reading the roofline numbers against: device-to-device copy and
read-only reduction of a buffer far larger than the caches (torch kernels; GB/s counts bytes read + written)."""
import time

import torch


def main():
    n = 1 << 30  # 4 GiB of float32
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for name, fn, moved in (("copy", lambda: b.copy_(a), 8 * n), ("read (sum)", lambda: a.sum(), 4 * n)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("%-10s %.0f GB/s" % (name, moved / dt / 1e9))


if __name__ == "__main__":
    main()
