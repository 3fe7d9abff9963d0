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
# the bubbles of the steady state: every gap above 15 us between the third and the last one-launch rollout of the trace
ra = [i for i, e in enumerate(ev) if "rollout_all_kernel" in e[2]]
if len(ra) >= 4:
    win = ev[ra[2]:ra[-1] + 1]
    its = len(ra) - 3
    print("iterations %d, %.3f ms each; gaps > 15 us inside them:" % (its, (win[-1][0] - win[0][0]) / 1e6 / its))
    tot, busy_to, last = 0, win[0][1], win[0][2]        # idle = time covered by NO kernel (copies may overlap a long kernel)
    for s1, e1, n1 in win[1:]:
        if s1 - busy_to > 15000:
            print("  %8.1f us : %s -> %s" % ((s1 - busy_to) / 1e3, short(last), short(n1)))
        if s1 > busy_to: tot += s1 - busy_to
        if e1 > busy_to: busy_to, last = e1, n1
    print("  idle %.1f us per iteration (all gaps)" % (tot / 1e3 / its))
