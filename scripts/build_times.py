"""Where the time of a commit (rtgi_hip_build on the 2^20-triangle soup) goes: run on the GPU box with
RTGI_HIP_BUILD_TIMES=1 for the library's own breakdown."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import scenes, tracer  # noqa: E402

pos, tri, _ = scenes.soup(1 << 20, seed=1)
for rep in range(3):
    t = tracer.HipBvhTracer(layout="flat", device=0)
    t0 = time.time()
    t.build(pos, tri)
    print("build call %.3f s" % (time.time() - t0), flush=True)
    t.close()
