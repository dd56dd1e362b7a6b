"""Median kernel duration of each consecutive run of 12 identical gemm launches in a kernel trace."""
import csv, glob, sys
f = sys.argv[1]
rows = [r for r in csv.DictReader(open(f)) if "gemm_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i in range(0, len(rows), 12):
    grp = rows[i:i + 12]
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp)
    r = grp[0]
    kind = "nt" if "gemm_nt" in r["Kernel_Name"] else "tn"
    print(f"{kind} grid=({int(r['Grid_Size_X'])//256},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) n={len(grp)} med={d[len(d)//2]:.1f} us min={d[0]:.1f}")
