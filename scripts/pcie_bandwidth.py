#!/usr/bin/env python3
"""Host <-> GPU copy bandwidth on this box from page-locked and from ordinary memory (torch copies), for sizes of
the host-buffer entry points' chunks. Context for the host-buffer rate in bench.py."""
import time

import torch

for mb in (8, 16, 48, 128):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, h in (("page-locked", torch.empty(n, dtype=torch.uint8).pin_memory()), ("ordinary", torch.empty(n, dtype=torch.uint8))):
        for what, src, dst in (("up", h, d), ("down", d, h)):
            dst.copy_(src)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            print("%4d MiB %-11s %-4s %6.1f GB/s" % (mb, name, what, 5 * n / (time.perf_counter() - t0) / 1e9))
