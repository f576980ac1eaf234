"""MFMA-pipe utilisation per kernel from a rocprofv3 counter_collection.csv holding SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE: busy / (GUI_ACTIVE per XCD x 1024 SIMDs).  usage: mfma_util.py <counter_collection.csv> [name filter]"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, c in agg.items():
    if flt not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in c: continue
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])   # summed over the 8 XCDs by rocprofv3
    mops = sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", [0])) / max(len(c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", [0])), 1)
    print(f"{k[:70]}: launches {len(c['GRBM_GUI_ACTIVE'])}  MFMA busy {busy:.4e}  GUI_ACTIVE {gui:.4e}  MOPS_F64 {mops:.4e}  -> MfmaUtil {100*busy/(gui/8*1024):.1f} %")
