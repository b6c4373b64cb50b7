#!/usr/bin/env python3
"""Dev tool: per-kernel sums of the counters of one rocprofv3 --pmc pass (counter_collection.csv)."""
import csv, glob, os, re, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else "conv3d"
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("ivf::", "")
    if pat not in k: continue
    key = (k, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", kv[1].get("SQ_WAVE_CYCLES", 0))):
    L = len(n[key])
    parts = [f"{k}={v / L:.3g}" for k, v in sorted(c.items())]
    extra = ""
    if c.get("SQ_LDS_IDX_ACTIVE"): extra += f" lds_conflict={c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}"
    if c.get("GRBM_GUI_ACTIVE"): extra += f" mfma_util={c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}"
    if c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        extra += " parked=%.2f stall=%.2f issue=%.2f lds_stall=%.2f" % (c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w, c.get("SQ_WAIT_INST_LDS", 0) / w)
    print(f"x{L:3d} grid {key[1]:>9s} {key[0][:60]:60s}{extra}\n      " + " ".join(parts))
