#!/usr/bin/env python3
"""Idle time between consecutive kernels of the optimizer-step loop from a rocprofv3 kernel trace: python tools/gaps_mlp.py <kernel_trace.csv>"""
import collections, csv, re, sys
def short(name):
    m = re.search(r"(fly_kernel<\d+>|mlp_\w+_kernel|ppo_\w+_kernel|rollout_\w+_kernel|dqn_\w+_kernel)", name)
    return m.group(1) if m else name.split("<")[0][-30:]
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
gaps, cnt, dur, dc = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    a, b = short(n0), short(n1)
    dur[a] += e0 - s0; dc[a] += 1
    if "mlp_" in a or "mlp_" in b or "rollout" in a:
        g = s1 - e0
        if g < 200000: gaps[(a, b)] += g; cnt[(a, b)] += 1
for k, g in sorted(gaps.items(), key=lambda kv: -cnt[kv[0]])[:14]:
    print("%7.2f us avg gap x %5d : %s -> %s" % (g / 1e3 / cnt[k], cnt[k], k[0], k[1]))
for k in dur:
    if "mlp_" in k or "rollout" in k or "gae" in k: print("%8.2f us avg x %5d : %s" % (dur[k] / 1e3 / dc[k], dc[k], k))
