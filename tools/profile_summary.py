#!/usr/bin/env python3
"""Condense rocprofv3 CSV output of tools/profile.sh (kernel-trace stats + separate --pmc passes of bench.py) into
profiles/<tag>_summary.md and profiles/<tag>_traffic.json (what bench.py's `roofline.traffic` / `roofline_shade` read).
Usage: tools/profile_summary.py <gpurun_out/prof_<tag> dir> <tag>"""
import glob
import json
import os
import re
import sys

import pandas as pd


def short(n):
    n = re.sub(r"pt::\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def find(src, sub, name):
    f = glob.glob(os.path.join(src, sub, "**", name), recursive=True)
    return f[0] if f else None


def bench_line(path):
    out = None
    if os.path.exists(path):
        for ln in open(path):
            if ln.startswith("{"):
                out = json.loads(ln)
    return out


def main(src, tag):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    args = open(os.path.join(src, "args.txt")).read().strip() if os.path.exists(os.path.join(src, "args.txt")) else ""
    lines = [f"# rocprofv3 summary `{tag}` (MI355X, gfx950)", ""]
    kt_bench = bench_line(os.path.join(src, "kt.log"))
    if kt_bench:
        lines += [f"Workload: {kt_bench['config']['workload']} — {kt_bench['ms_per_step']:.2f} ms per step under the kernel trace, "
                  f"{kt_bench['value'] / 1e3:.2f} Gray/s traversed ({kt_bench['config']['cast_Mray_per_s'] / 1e3:.2f} Gray/s at the reference's call sites).", ""]
    ksf = find(src, "kt", "*kernel_stats.csv")
    ks = pd.read_csv(ksf)
    ks["kernel"] = ks["Name"].map(short)
    lines += [f"## `rocprofv3 --kernel-trace --stats -- python bench.py {args} --pipelines 1 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-ms` (two steps)", "",
              "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for _, r in ks.iterrows():
        lines.append(f"| {r['kernel']} | {r['Calls']} | {r['TotalDurationNs'] / 1e6:.2f} | {r['AverageNs'] / 1e3:.1f} | {r['Percentage']:.2f} |")
    lines += ["", "`k_closest<.., 1, ..>` (the BSDF-sampled NEE rays, about 1 % of them since the shading pass culls the rest): where it is a launch of its own it runs on a side "
              "stream beside `k_any` and its duration mostly measures waiting for wave slots, so the column sums can exceed the wall time.  `k_shade_surface<1u, false, true>` is the "
              "Lambertian pass that walks its own shadow rays (LDS-resident scenes, round 4): no `k_any` row then, unless GGX surfaces queue theirs."]
    kt_ms = {r["kernel"]: r["TotalDurationNs"] / 1e6 for _, r in ks.iterrows()}
    kt_steps = 2.0
    traffic = {}
    frames = []
    pmc_bench = None
    for d in sorted(os.listdir(src)):
        f = find(src, d, "*counter_collection.csv") if d.startswith("pmc") and os.path.isdir(os.path.join(src, d)) else None
        if f:
            frames.append(pd.read_csv(f))
        if d.startswith("pmc") and d.endswith(".log"):
            pmc_bench = bench_line(os.path.join(src, d)) or pmc_bench
    g = None
    if frames:
        pm = pd.concat(frames)
        pm["kernel"] = pm["Kernel_Name"].map(short)
        g = pm.groupby(["kernel", "Counter_Name"])["Counter_Value"].sum().unstack()
        n = pm.groupby(["kernel", "Counter_Name"])["Dispatch_Id"].nunique().unstack()
        spp_p = pmc_bench["config"]["workload"].split(" spp")[0].split(", ")[-1] if pmc_bench else "?"
        lines += ["", f"## PMC passes (`--pmc`, one pass per counter group; bench.py {args} --pipelines 1 --steps 1 --warmup 0, {spp_p} spp)", "",
                  "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads "
                  "(MI355X_MICROARCH.md §HBM), so read bytes below = 2 x FETCH_SIZE x 1024.", ""]
        cols = [c for c in ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
                            "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE",
                            "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"] if c in g.columns]
        lines += ["| kernel | " + " | ".join(cols) + " |", "|---|" + "---|" * len(cols)]
        for k, r in g.iterrows():
            lines.append(f"| {k} | " + " | ".join(f"{r[c]:.4g}" if pd.notna(r[c]) else "" for c in cols) + " |")
        if "SQ_ACTIVE_INST_VALU" in g.columns and "SQ_WAVE_CYCLES" in g.columns:
            lines += ["", "| kernel | VALU-active share of wave lifetime | lanes per VALU instruction | VALU wave-instr per wave | L2 hit rate |", "|---|---|---|---|---|"]
            for k, r in g.iterrows():
                if pd.notna(r.get("SQ_WAVE_CYCLES")) and r["SQ_WAVE_CYCLES"] > 0:
                    l2 = f"{r['TCC_HIT_sum'] / max(r['TCC_HIT_sum'] + r['TCC_MISS_sum'], 1):.2f}" if "TCC_HIT_sum" in g.columns and pd.notna(r.get("TCC_HIT_sum")) else ""
                    lines.append(f"| {k} | {r['SQ_ACTIVE_INST_VALU'] / r['SQ_WAVE_CYCLES']:.3f} | {r['SQ_THREAD_CYCLES_VALU'] / max(r['SQ_ACTIVE_INST_VALU'], 1):.1f} | "
                                 f"{r['SQ_INSTS_VALU'] / max(r['SQ_WAVES'], 1):.0f} | {l2} |")
        for k, r in g.iterrows():
            if "FETCH_SIZE" in g.columns and "WRITE_SIZE" in g.columns and pd.notna(r.get("FETCH_SIZE")) and pd.notna(r.get("WRITE_SIZE")):
                launches = int(n.loc[k, "FETCH_SIZE"])
                traffic[k] = {"launches": launches, "read_bytes_per_launch": 2 * r["FETCH_SIZE"] * 1024 / launches,
                              "write_bytes_per_launch": r["WRITE_SIZE"] * 1024 / launches}
                traffic[k]["hbm_bytes_per_launch"] = traffic[k]["read_bytes_per_launch"] + traffic[k]["write_bytes_per_launch"]
    # the dominant kernel is k_closest in its two world modes (PRIMARY = bounce 0, WORLD = later bounces): bytes per traced ray
    try:
        rays = pmc_bench["roofline"]["rays_per_launch"] * pmc_bench["roofline"]["launches"]  # rays that reached the kernel in the profiled (timed) step
        spp_p = int(pmc_bench["config"]["workload"].split(" spp")[0].split(", ")[-1])
        # world closest-hit launches: k_closest<.., WORLD = 0 | PRIMARY = 3, ..> and k_trace_fused (bounces >= 1 of LDS scenes: world closest hit + the few BSDF-sampled NEE rays)
        keys = [k for k in traffic if re.match(r"k_closest\d?<\w+, [03](, \w+)*>$", k) or k.startswith("k_trace_fused")]
        tot = sum(traffic[k]["hbm_bytes_per_launch"] * traffic[k]["launches"] for k in keys)
        # resident waves per SIMD of the traversal kernels: 5 (92-96 VGPRs under __launch_bounds__(256, 5) since round 4, BVH in LDS or in global memory; rounds 1-3: 4)
        waves = 5
        va = sum(g.loc[k, "SQ_ACTIVE_INST_VALU"] for k in keys) / sum(g.loc[k, "SQ_WAVE_CYCLES"] for k in keys)
        ln = sum(g.loc[k, "SQ_THREAD_CYCLES_VALU"] for k in keys) / sum(g.loc[k, "SQ_ACTIVE_INST_VALU"] for k in keys)
        traffic["k_closest_world"] = {"kernels": keys, "rays": rays, "hbm_bytes": tot, "hbm_bytes_per_ray": tot / rays, "spp": spp_p,
                                      "valu_active_frac": va, "lanes_per_valu_instr": ln, "waves_per_simd": waves,
                                      "algorithmic_bytes_per_ray": pmc_bench["roofline"].get("algorithmic_bytes_per_ray"),
                                      "note": "FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as read; valu_active_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of a "
                                              "wave's lifetime it issues VALU; x waves_per_simd = VALU busy: 5 waves per SIMD resident since round 4), "
                                              "lanes_per_valu_instr = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64)"}
        lines += ["", f"k_closest PRIMARY+WORLD: VALU-active share of wave lifetime {va:.3f} (x {waves} resident waves per SIMD = {waves * va:.2f} VALU busy); lanes active per VALU instruction {ln:.1f} of 64 "
                  f"(effective VALU use {waves * va * ln / 64:.2f}).",
                  "", f"Dominant kernel (k_closest, modes PRIMARY+WORLD): {tot / 1e9:.2f} GB of HBM traffic for {rays / 1e6:.1f} M rays = **{tot / rays:.1f} B/ray** by the counters."]
        # the memory-bound kernel: all shading launches of the profiled step
        sk = [k for k in traffic if k.startswith("k_shade_surface")]
        sb = sum(traffic[k]["hbm_bytes_per_launch"] * traffic[k]["launches"] for k in sk)
        s_ms = sum(v for k, v in kt_ms.items() if k.startswith("k_shade_surface")) / kt_steps
        spp_kt = int(kt_bench["config"]["workload"].split(" spp")[0].split(", ")[-1]) if kt_bench else spp_p
        traffic["k_shade_surface"] = {"kernels": sk, "hbm_bytes": sb, "spp": spp_p, "hbm_bytes_per_spp": sb / spp_p,
                                      "kernel_trace_ms_per_step": s_ms, "kernel_trace_spp": spp_kt,
                                      "GBps": (sb / spp_p * spp_kt) / (s_ms * 1e-3) / 1e9 if s_ms > 0 else None}
        if s_ms > 0:
            gbps = traffic["k_shade_surface"]["GBps"]
            inl = any(k.endswith(", true>") for k in sk)
            lines += ["", f"Shading pass (k_shade_surface, all launches{'; the Lambertian launches include their shadow walk, which moves no bytes' if inl else ''}): "
                      f"{sb / 1e9:.2f} GB per {spp_p}-spp step by the counters = {sb / spp_p / 1e6:.1f} MB per spp; "
                      f"{s_ms:.2f} ms per {spp_kt}-spp step in the kernel trace -> **{gbps:.0f} GB/s = {gbps / 8000:.2f} of the 8 TB/s peak**."]
    except Exception as e:  # noqa: BLE001
        lines += ["", f"(bytes per ray not derived: {e})"]
    open(os.path.join(out_dir, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(os.path.join(out_dir, f"{tag}_traffic.json"), "w"), indent=1)
    if ksf:
        import shutil
        shutil.copy(ksf, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
