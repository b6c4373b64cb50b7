#!/usr/bin/env python3
"""Where the waves of each kernel spend their cycles (dev tool): one rocprofv3 --pmc pass with
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES.  MI355X_MICROARCH.md (rocprofv3 PMC slots): WAIT_ANY = parked on
s_waitcnt / barrier, WAIT_INST_ANY = issue stall (MFMA dependency / pipe busy), ACTIVE_INST_ANY = issuing;
the three are disjoint and sum to about WAVE_CYCLES."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"\s+", "", name.replace("ivf::", ""))


f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
first = next(int(r["Dispatch_Id"]) for r in rows if "mask_reg_kernel" in r["Kernel_Name"])
acc = defaultdict(lambda: defaultdict(float))
for r in rows:
    if int(r["Dispatch_Id"]) >= first:
        acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, c in acc.items():
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    if w <= 0:
        continue
    out[k] = {n[3:].lower() + "_share": c.get(n, 0.0) / w for n in
              ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
               "SQ_ACTIVE_INST_LDS")}
    out[k]["wave_cycles"] = w
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["wave_cycles"]))
json.dump({"how": __doc__, "kernels": out}, open(sys.argv[2], "w"), indent=1)
print("share of wave cycles:  parked(waitcnt/barrier)  issue-stall  issuing | of which LDS-issue-stall, VALU issuing, LDS issuing")
for k, e in list(out.items())[:22]:
    print(f"{e['wait_any_share']*100:5.1f} {e['wait_inst_any_share']*100:5.1f} {e['active_inst_any_share']*100:5.1f} | "
          f"{e['wait_inst_lds_share']*100:5.1f} {e['active_inst_valu_share']*100:5.1f} {e['active_inst_lds_share']*100:5.1f}  {k}")
