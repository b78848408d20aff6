#!/usr/bin/env python3
"""Writes tests/golden/hit_vectors.npz: for every scene of render-data/ and two tree settings, seeded rays and
the hit records, any-hit bytes and per-ray counts that this repository's CPU oracle gives for them.

    python tests/golden/make_hit_vectors.py

The vectors are NOT outputs of the reference renderer (it cannot be built in this image): they freeze the
oracle's answers at the time it was checked against the reference's own known-answer tests, so that a later
change of the oracle or of the kernels shows up against stored numbers and not only against each other.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as orc  # noqa: E402
from tests.util import HIT_VECTOR_CASES, hit_vector_rays, obj_scene  # noqa: E402


def main():
    out = {}
    for k, (name, layout, split, planes) in enumerate(HIT_VECTOR_CASES):
        pos, tri, _ = obj_scene(name)
        rays = hit_vector_rays(pos, k)
        o = orc.Bbvh(pos, tri, layout=layout, split=split, planes=planes)
        hits, counts = o.closest_hit(rays, counts=True)
        occluded, any_counts = o.any_hit(rays, counts=True)
        key = "%s/%s/%s" % (name, layout, split)
        out[key + "/hits"] = hits.view(np.uint32).reshape(-1, 4)  # bit patterns of t, beta, gamma; ref
        out[key + "/counts"] = counts[:, 0:2].astype(np.uint32)
        out[key + "/occluded"] = occluded
        out[key + "/any_counts"] = any_counts[:, 0:2].astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, "hit_vectors.npz"), **out)
    print("wrote %d arrays" % len(out))


if __name__ == "__main__":
    main()
