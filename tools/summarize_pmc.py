#!/usr/bin/env python3
"""Turn rocprofv3 outputs (kernel-trace stats + --pmc passes of tools/prof_kernels.py) into the
small files kept under profiles/: a per-kernel CSV of counter means and a JSON with the HBM bytes
per launch that bench.py reports as `roofline.traffic`.

    python tools/summarize_pmc.py <tag> <kernel_stats_dir> <pmc_dir> [<pmc_dir> ...]

HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so reads of the
16-byte-per-lane streaming kernels are doubled (kernels listed in WIDE_READS); WRITE_SIZE is exact
for 16-byte-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

WIDE_READS = ("mlp_fused_step_kernel", "mlp_fused_step_h2_kernel", "mlp_grad_reduce_h2_kernel", "mlp_forward_kernel", "mlp_backward_dx_kernel", "mlp_fwd_bwd_kernel", "mlp_grad_w_kernel", "mlp_grad_w_b3_kernel",
              "mlp_grad_reduce_kernel", "dqn_td_kernel", "dqn_grad_w_kernel", "dqn_grad_reduce_kernel", "dqn_forward_kernel", "dqn_act_kernel", "dqn_chain_kernel", "dqn_dw2_kernel", "dqn_chain_h2_kernel", "dqn_dw2_h2_kernel", "dqn_dw2r_h2_kernel")


def short(name):
    m = re.search(r"(fly_kernel<\d+>|mlp_\w+_kernel|ppo_\w+_kernel|dqn_\w+_kernel|rollout_step_kernel|rollout_all_fs_kernel|rollout_all_kernel)", name)
    return m.group(1) if m else None


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    rows = []
    for f in glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    with open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "percent"])
        for r in rows:
            w.writerow([r["Name"][:100], r["Calls"], "%.3f" % (float(r["AverageNs"]) / 1e3), "%.3f" % (float(r["MinNs"]) / 1e3),
                        "%.3f" % (float(r["MaxNs"]) / 1e3), "%.3f" % (float(r["TotalDurationNs"]) / 1e6), r["Percentage"]])
    agg = collections.defaultdict(list)
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    agg[(k, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(out_dir, "%s_pmc_summary.csv" % tag), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "grid_threads", "counter", "dispatches", "mean_per_dispatch"])
        for (k, g, c), v in sorted(agg.items()):
            w.writerow([k, g, c, len(v), "%.6g" % (sum(v) / len(v))])
    traffic = {}
    for (k, g, c), v in agg.items():
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            t = traffic.setdefault("%s@%d" % (k, g), {"fetch_kib": 0.0, "write_kib": 0.0})
            t["fetch_kib" if c == "FETCH_SIZE" else "write_kib"] = sum(v) / len(v)
    for key, t in traffic.items():
        mult = 2.0 if key.split("@")[0] in WIDE_READS else 1.0
        t["read_correction"] = mult
        t["hbm_bytes_per_launch"] = int((t["fetch_kib"] * mult + t["write_kib"]) * 1024)
        if key.split("@")[0] in ("dqn_chain_kernel", "dqn_dw2_kernel", "dqn_chain_h2_kernel", "dqn_dw2_h2_kernel", "dqn_dw2r_h2_kernel"):     # one launch per UPDATE: prof_kernels.py's PROF_DQN_MB sampled steps
            t["sampled_steps_per_launch"] = int(os.environ.get("PROF_DQN_MB", "4"))
            t["hbm_bytes_per_sampled_step"] = t["hbm_bytes_per_launch"] // t["sampled_steps_per_launch"]
    with open(os.path.join(out_dir, "%s_traffic.json" % tag), "w") as fo:
        json.dump(traffic, fo, indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1, sort_keys=True))
    # vector instructions per wave of the env kernels (bench.py's `valu_roofline` reads the newest *_valu.json)
    valu = {}
    bench_grid = int(os.environ.get("PROF_ENVS", "8192")) // 32 * 256        # the entry of the bench size wins over other sizes
    for (k, g, c), v in sorted(agg.items(), key=lambda kv: kv[0][1] == bench_grid):
        if c == "SQ_INSTS_VALU" and (k.startswith("fly_kernel<63>") or k.startswith("rollout_all")):
            waves = agg.get((k, g, "SQ_WAVES"))
            if waves:
                w = sum(waves) / len(waves)
                e = {"valu_insts_per_wave": int(round(sum(v) / len(v) / w)), "sq_insts_valu_per_launch": sum(v) / len(v), "waves": int(w),
                     "grid_threads": g}
                for extra in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"):
                    x = agg.get((k, g, extra))
                    if x:
                        e[extra.lower() + "_per_launch"] = sum(x) / len(x)
                valu[k] = e
    if valu:
        steps = int(os.environ.get("PROF_T", "80"))
        for k, e in valu.items():
            if k.startswith("rollout_all"):
                e["env_steps_per_launch"] = steps
                e["valu_insts_per_wave_per_env_step"] = int(round(e["valu_insts_per_wave"] / steps))
        valu["clock_ghz"] = 2.1
        valu["source"] = "rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -- python3 tools/prof_kernels.py (tools/collect_profiles.sh %s)" % tag
        with open(os.path.join(out_dir, "%s_valu.json" % tag), "w") as fo:
            json.dump(valu, fo, indent=1, sort_keys=True)
        print(json.dumps(valu, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
