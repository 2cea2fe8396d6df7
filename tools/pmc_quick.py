#!/usr/bin/env python3
"""Per-kernel sums of the counters tools/pmc_quick.sh collected: pmc_quick.py <dir>"""
import glob, os, re, sys
import pandas as pd
fr = [pd.read_csv(f) for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True)]
pm = pd.concat(fr)
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", re.sub(r"pt::\(anonymous namespace\)::", "", n)))
pm["kernel"] = pm["Kernel_Name"].map(short)
g = pm.groupby(["kernel", "Counter_Name"])["Counter_Value"].sum().unstack()
g = g[g.index.str.startswith("k_")]
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
out = pd.DataFrame(index=g.index)
out["waves"] = g["SQ_WAVES"]
out["VALU_M"] = g["SQ_INSTS_VALU"] / 1e6
out["SALU_M"] = g.get("SQ_INSTS_SALU", 0) / 1e6
out["LDS_M"] = g.get("SQ_INSTS_LDS", 0) / 1e6
out["lanes/VALU"] = g["SQ_THREAD_CYCLES_VALU"] / g["SQ_ACTIVE_INST_VALU"] / 4.0 * 4.0
out["valu_active"] = g["SQ_ACTIVE_INST_VALU"] / g["SQ_WAVE_CYCLES"]
out["cyc/VALU"] = g["SQ_ACTIVE_INST_VALU"] / g["SQ_INSTS_VALU"]
out["wave_Mcyc"] = g["SQ_WAVE_CYCLES"] / 1e6
out["busy_Mcyc"] = g["SQ_BUSY_CYCLES"] / 1e6
if "SQ_INST_CYCLES_SALU" in g: out["salu_cyc_frac"] = g["SQ_INST_CYCLES_SALU"] / g["SQ_WAVE_CYCLES"]
if "SQ_WAIT_INST_ANY" in g: out["wait_inst_any"] = g["SQ_WAIT_INST_ANY"] / g["SQ_WAVE_CYCLES"]
if "SQ_ACTIVE_INST_ANY" in g: out["active_any"] = g["SQ_ACTIVE_INST_ANY"] / g["SQ_WAVE_CYCLES"]
print(out.round(3).to_string())
