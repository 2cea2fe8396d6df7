#!/usr/bin/env python3
"""Condense tools/profile_scene.sh output into profiles/<tag>_summary.md (+ <tag>_roofline.json).  Usage: scene_summary.py <dir> <tag>
one_frame.py renders the frame twice (warm-up + timed), so every count below is over TWO frames."""
import json, os, re, sys
import pandas as pd

def short(n):
    n = re.sub(r"pt::\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)

src, tag = sys.argv[1], sys.argv[2]
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
roof = None
try:
    roof = json.loads(open(os.path.join(src, "roofline.json")).read().strip().split("\n")[-1])
except Exception:
    pass
lines = [f"# rocprofv3 summary `{tag}` (MI355X, gfx950)", ""]
if roof:
    lines += [f"Scene `{roof['scene']}`: {roof['triangles']} triangles, BVH blob {roof['scene_bytes'] / 1e6:.2f} MB (LDS-resident: {bool(roof['lds_scene'])}), "
              f"1920x1080, {roof['spp']} spp, depth {roof['depth']}: **{roof['frame_ms']:.2f} ms per frame, {roof['Mray_per_s'] / 1e3:.2f} Gray/s**.", "",
              f"World closest-hit kernel: {roof['rays_closest'] / 1e6:.1f} M rays in {roof['k_closest_ms']:.2f} ms over {roof['k_closest_launches']} launches "
              f"({roof['k_closest_Mray_per_s'] / 1e3:.2f} Gray/s in-kernel); the oracle visits {roof['nodes_visited_per_closest_ray']:.1f} nodes and tests "
              f"{roof['triangles_tested_per_closest_ray']:.1f} triangles per ray on the same rays, i.e. **{roof['algorithmic_bytes_per_closest_ray']:.0f} algorithmic "
              f"bytes per ray** (48 + 32 per node + 48 per triangle, SURVEY 8d) = {roof['roofline']['achieved']:.0f} GB/s = "
              f"**{roof['roofline']['frac']:.3f} of the 8 TB/s HBM roofline**.", ""]
ks = pd.read_csv(os.path.join(src, "kt", "r1_kernel_stats.csv"))
ks["kernel"] = ks["Name"].map(short)
lines += ["## `rocprofv3 --kernel-trace --stats -- python tools/one_frame.py 1 0 <scene> <spp>` (two frames)", "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for _, r in ks.iterrows():
    if r["Percentage"] >= 0.05:
        lines.append(f"| {r['kernel']} | {r['Calls']} | {r['TotalDurationNs'] / 1e6:.2f} | {r['AverageNs'] / 1e3:.1f} | {r['Percentage']:.2f} |")
frames = []
for d in sorted(os.listdir(src)):
    f = os.path.join(src, d, "r1_counter_collection.csv")
    if d.startswith("pmc") and os.path.exists(f):
        frames.append(pd.read_csv(f))
summary = {"roofline": roof, "kernels": {}}
if frames:
    pm = pd.concat(frames)
    pm["kernel"] = pm["Kernel_Name"].map(short)
    g = pm.groupby(["kernel", "Counter_Name"])["Counter_Value"].sum().unstack()
    lines += ["", "## PMC passes (`--pmc`, one pass per group, never combined with other trace domains; two frames)", "",
              "FETCH_SIZE / WRITE_SIZE are KiB; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts wide coalesced reads at half, MI355X_MICROARCH.md HBM section; "
              "narrow BVH gathers are uncalibrated, so treat the read figure of the traversal kernels as an upper bound of about 2x).", ""]
    cols = [c for c in g.columns]
    lines += ["| kernel | " + " | ".join(cols) + " |", "|---|" + "---|" * len(cols)]
    for k, r in g.iterrows():
        if k.startswith("k_"):
            lines.append(f"| {k} | " + " | ".join(f"{r[c]:.4g}" if pd.notna(r[c]) else "" for c in cols) + " |")
    lines += ["", "| kernel | HBM read GB | HBM write GB | L2 hit rate | VALU-active share of wave lifetime | lanes active per VALU instruction | VALU wave-instr |", "|---|---|---|---|---|---|---|"]
    for k, r in g.iterrows():
        if not k.startswith("k_"):
            continue
        def v(c): return r[c] if c in g.columns and pd.notna(r[c]) else None
        rd = 2 * v("FETCH_SIZE") * 1024 / 1e9 if v("FETCH_SIZE") is not None else None
        wr = v("WRITE_SIZE") * 1024 / 1e9 if v("WRITE_SIZE") is not None else None
        hit = v("TCC_HIT_sum") / (v("TCC_HIT_sum") + v("TCC_MISS_sum")) if v("TCC_HIT_sum") is not None and v("TCC_MISS_sum") is not None and (v("TCC_HIT_sum") + v("TCC_MISS_sum")) > 0 else None
        act = v("SQ_ACTIVE_INST_VALU") / v("SQ_WAVE_CYCLES") if v("SQ_ACTIVE_INST_VALU") is not None and v("SQ_WAVE_CYCLES") else None
        lanes = v("SQ_THREAD_CYCLES_VALU") / v("SQ_ACTIVE_INST_VALU") if v("SQ_THREAD_CYCLES_VALU") is not None and v("SQ_ACTIVE_INST_VALU") else None
        summary["kernels"][k] = {"hbm_read_GB": rd, "hbm_write_GB": wr, "l2_hit_rate": hit, "valu_active_frac": act, "lanes_per_valu_instr": lanes, "valu_insts": v("SQ_INSTS_VALU")}
        f = lambda x, p=3: "" if x is None else f"{x:.{p}g}"
        lines.append(f"| {k} | {f(rd)} | {f(wr)} | {f(hit)} | {f(act)} | {f(lanes)} | {f(v('SQ_INSTS_VALU'), 4)} |")
    if roof:
        keys = [k for k in summary["kernels"] if re.match(r"k_closest<\w+, [03](, \w+)?>$", k)]
        tot = sum((summary["kernels"][k]["hbm_read_GB"] or 0) + (summary["kernels"][k]["hbm_write_GB"] or 0) for k in keys) * 1e9 / 2  # two frames
        alg = roof["algorithmic_bytes_per_closest_ray"] * roof["rays_closest"]
        summary["k_closest_world"] = {"kernels": keys, "hbm_bytes_per_frame": tot, "hbm_bytes_per_ray": tot / max(roof["rays_closest"], 1), "algorithmic_bytes_per_frame": alg,
                                      "counter_over_algorithmic": tot / max(alg, 1)}
        lines += ["", f"World closest-hit kernels ({', '.join(keys)}): {tot / 1e9:.2f} GB of counted HBM traffic per frame = {tot / max(roof['rays_closest'], 1):.0f} B/ray against "
                  f"{roof['algorithmic_bytes_per_closest_ray']:.0f} algorithmic B/ray: **counter / algorithmic = {tot / max(alg, 1):.2f}** "
                  "(below 1: node and triangle reads are served by L2 / Infinity Cache, the scene is far smaller than either)."]
open(os.path.join(out_dir, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out_dir, f"{tag}_roofline.json"), "w"), indent=1)
print("\n".join(lines))
