#!/usr/bin/env python3
"""MFMA utilisation and LDS conflict share per kernel from one rocprofv3 --pmc pass (dev tool).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
              --kernel-trace --output-format csv -d D -- python3 bench.py ...
    python tools/pmc_mfma.py D out.json

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs).  MFMA_BUSY is summed over
all SIMDs and counts 32 cycles per v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md; checked: the stem
forward kernel issues 236 M MFMAs per launch at B=128 and reads 7.553e9); GRBM_GUI_ACTIVE comes back summed
over the 8 XCDs (141 M "cycles" for an 8.0 ms launch = 8 x 2.2 GHz), hence the division by 8.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"\s+", "", name.replace("ivf::", ""))


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    first = next(int(r["Dispatch_Id"]) for r in rows if "mask_reg_kernel" in r["Kernel_Name"])
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    dur = defaultdict(dict)
    inst = defaultdict(int)   # rows per (dispatch, counter): >1 means one row per counter instance
    for r in rows:
        if int(r["Dispatch_Id"]) < first:
            continue
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
        dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        inst[(r["Dispatch_Id"], r["Counter_Name"])] += 1
    print("rows per (dispatch, counter):", sorted(set(inst.values())))
    out = {}
    for k, c in acc.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0:
            continue
        e = {"launches": len(n[k]), "gui_active_cycles_per_launch": gui / len(n[k]),
             "ns_per_launch": sum(dur[k].values()) / len(n[k]),
             "mfma_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / len(n[k]),
             "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0)}
        if c.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
            e["lds_bank_conflict_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        out[k] = e
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["gui_active_cycles_per_launch"] * kv[1]["launches"]))
    json.dump({"how": __doc__, "kernels": out}, open(sys.argv[2], "w"), indent=1)
    for k, e in list(out.items())[:20]:
        print(f"mfma_util {e['mfma_util']*100:5.1f} %  lds_conflict {e.get('lds_bank_conflict_share', 0)*100:5.1f} %  "
              f"x{e['launches']:4d}  {k}")


if __name__ == "__main__":
    main()
