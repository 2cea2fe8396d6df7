#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace stats + separate --pmc passes) into profiles/<tag>_summary.md and
profiles/<tag>_traffic.json.  Usage: tools/profile_summary.py <gpurun_out/prof dir> <tag>"""
import json
import os
import re
import sys

import pandas as pd


def short(n):
    n = re.sub(r"pt::\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def main(src, tag):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    lines = [f"# rocprofv3 summary `{tag}` (MI355X, gfx950)", ""]
    ks = pd.read_csv(os.path.join(src, "kt", "r1_kernel_stats.csv"))
    ks["kernel"] = ks["Name"].map(short)
    lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline`", "",
              "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for _, r in ks.iterrows():
        lines.append(f"| {r['kernel']} | {r['Calls']} | {r['TotalDurationNs'] / 1e6:.2f} | {r['AverageNs'] / 1e3:.1f} | {r['Percentage']:.2f} |")
    lines += ["", "`k_closest<.., 1, ..>` (the BSDF-sampled NEE rays, about 1 % of them since the shading pass culls the rest) is launched on a side stream "
              "beside `k_any`: its duration overlaps `k_any`'s and mostly measures waiting for wave slots, so the column sums exceed the wall time."]
    traffic = {}
    frames = []
    for d in sorted(os.listdir(src)):
        f = os.path.join(src, d, "r1_counter_collection.csv")
        if d.startswith("pmc") and os.path.exists(f):
            frames.append(pd.read_csv(f))
    if frames:
        pm = pd.concat(frames)
        pm["kernel"] = pm["Kernel_Name"].map(short)
        g = pm.groupby(["kernel", "Counter_Name"])["Counter_Value"].sum().unstack()
        n = pm.groupby(["kernel", "Counter_Name"])["Dispatch_Id"].nunique().unstack()
        lines += ["", "## PMC passes (`--pmc`, one pass per counter group; bench.py --steps 1 --warmup 0 --spp 43)", "",
                  "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads "
                  "(MI355X_MICROARCH.md §HBM), so read bytes below = 2 x FETCH_SIZE x 1024.", ""]
        cols = [c for c in ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
                            "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"] if c in g.columns]
        lines += ["| kernel | " + " | ".join(cols) + " |", "|---|" + "---|" * len(cols)]
        for k, r in g.iterrows():
            lines.append(f"| {k} | " + " | ".join(f"{r[c]:.4g}" if pd.notna(r[c]) else "" for c in cols) + " |")
        if "SQ_ACTIVE_INST_VALU" in g.columns and "SQ_WAVE_CYCLES" in g.columns:
            lines += ["", "| kernel | VALU-active share of wave lifetime | VALU wave-instr per wave |", "|---|---|---|"]
            for k, r in g.iterrows():
                if pd.notna(r.get("SQ_WAVE_CYCLES")) and r["SQ_WAVE_CYCLES"] > 0:
                    lines.append(f"| {k} | {r['SQ_ACTIVE_INST_VALU'] / r['SQ_WAVE_CYCLES']:.3f} | {r['SQ_INSTS_VALU'] / max(r['SQ_WAVES'], 1):.0f} |")
        for k, r in g.iterrows():
            if "FETCH_SIZE" in g.columns and "WRITE_SIZE" in g.columns and pd.notna(r.get("FETCH_SIZE")) and pd.notna(r.get("WRITE_SIZE")):
                launches = int(n.loc[k, "FETCH_SIZE"])
                traffic[k] = {"launches": launches, "read_bytes_per_launch": 2 * r["FETCH_SIZE"] * 1024 / launches,
                              "write_bytes_per_launch": r["WRITE_SIZE"] * 1024 / launches}
                traffic[k]["hbm_bytes_per_launch"] = traffic[k]["read_bytes_per_launch"] + traffic[k]["write_bytes_per_launch"]
    # the dominant kernel is k_closest in its two world modes (PRIMARY = bounce 0, WORLD = later bounces): bytes per traced ray
    try:
        bench = None
        for name in sorted(os.listdir(src)):
            if name.startswith("pmc") and name.endswith(".log"):
                for ln in open(os.path.join(src, name)):
                    if ln.startswith("{"):
                        bench = json.loads(ln)
        rays = bench["roofline"]["rays_per_launch"] * bench["roofline"]["launches"]  # rays that reached the kernel in the profiled (timed) step
        keys = [k for k in traffic if re.match(r"k_closest<\w+, [03](, \w+)?>$", k)]
        tot = sum(traffic[k]["hbm_bytes_per_launch"] * traffic[k]["launches"] for k in keys)
        va = sum(g.loc[k, "SQ_ACTIVE_INST_VALU"] for k in keys) / sum(g.loc[k, "SQ_WAVE_CYCLES"] for k in keys) if "SQ_WAVE_CYCLES" in g.columns else None
        ln = sum(g.loc[k, "SQ_THREAD_CYCLES_VALU"] for k in keys) / sum(g.loc[k, "SQ_ACTIVE_INST_VALU"] for k in keys) if "SQ_THREAD_CYCLES_VALU" in g.columns else None
        traffic["k_closest_world"] = {"kernels": keys, "rays": rays, "hbm_bytes": tot, "hbm_bytes_per_ray": tot / rays, "spp": bench["config"]["workload"].split(" spp")[0].split(", ")[-1],
                                      "valu_active_frac": va, "lanes_per_valu_instr": ln,
                                      "note": "PMC passes at batch 43 spp; FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as read; valu_active_frac = "
                                              "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of a wave's lifetime it issues VALU; x resident waves per SIMD = VALU busy), "
                                              "lanes_per_valu_instr = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64)"}
        if va is not None: lines += ["", f"k_closest PRIMARY+WORLD: VALU-active share of wave lifetime {va:.3f}; lanes active per VALU instruction {ln if ln is None else round(ln, 1)} of 64."]
        lines += ["", f"Dominant kernel (k_closest, modes PRIMARY+WORLD): {tot / 1e9:.2f} GB of HBM traffic for {rays / 1e6:.1f} M rays = "
                  f"**{tot / rays:.1f} B/ray** (algorithmic: 48 B/ray)."]
    except Exception as e:  # noqa: BLE001
        lines += ["", f"(bytes per ray not derived: {e})"]
    open(os.path.join(out_dir, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(os.path.join(out_dir, f"{tag}_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
