"""Summarises rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch)."""
import csv, sys, collections, glob, os
def summarize(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
    for k, cs in acc.items():
        d = list(dur[k].values())
        print(f"kernel {k}: dispatches={len(d)} mean_ns={sum(x[0] for x in d)/len(d):.0f} vgpr={d[0][1]} sgpr={d[0][2]} lds={d[0][3]} grid={d[0][4]} wg={d[0][5]}")
        for c, v in sorted(cs.items()):
            print(f"    {c:28s} mean/dispatch = {sum(v)/len(v):.6g}  (n={len(v)})")
for p in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)):
        print("==", f)
        summarize(f)
