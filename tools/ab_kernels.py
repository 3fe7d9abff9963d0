#!/usr/bin/env python3
"""A/B timing of the hot kernels for the library named by FLYHIP_LIB (default: the in-tree build).
Prints avg launch microseconds at the bench sizes (HIP events, back-to-back launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ks = bench.kernel_rooflines(8192, 80, int(os.environ.get("REPS", "100")))
print(os.environ.get("FLYHIP_LIB", "in-tree"), " | ".join("%s %.1f" % (k["kernel"].split()[0], k["avg_launch_us"]) for k in ks))
