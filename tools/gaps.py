#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 kernel trace (csv): python tools/gaps.py <kernel_trace.csv>
Prints busy/idle totals over the steady part of the run and the largest idle-gap sources (previous kernel -> next)."""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(fly_kernel<\d+>|mlp_\w+_kernel(<[^>]*>)?|ppo_\w+_kernel|dqn_\w+_kernel)", name)
    return m.group(1) if m else name.split("<")[0][-36:]

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[len(ev) // 3:]                      # skip warm-up
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.Counter()
cnt = collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    g = max(0, s1 - e0)
    key = (short(n0), short(n1))
    gaps[key] += g
    cnt[key] += 1
print("span %.2f ms, kernels busy %.2f ms (%.1f %%), idle %.2f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
for k, g in gaps.most_common(12):
    print("%8.3f ms total, %6.2f us avg x %5d : %s -> %s" % (g / 1e6, g / 1e3 / cnt[k], cnt[k], k[0], k[1]))
