"""Idle time between consecutive kernels of a rocprofv3 kernel trace: total, histogram, and the largest gaps
with the kernels on either side (finds host-bound stretches and the matcher bubble)."""
import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last ~40 % of the trace (steady-state steps)
t0 = int(rows[0]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
cut = t0 + int((t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = []
end = int(rows[0]["End_Timestamp"])
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - end
    gaps.append((g, a["Kernel_Name"][:60], b["Kernel_Name"][:60], int(b["Start_Timestamp"]) - int(rows[0]["Start_Timestamp"])))
    end = max(end, int(b["End_Timestamp"]))
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms")
h = collections.Counter()
for g, *_ in gaps:
    k = "<0" if g < 0 else "<1us" if g < 1000 else "<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"
    h[k] += max(g, 0)
for k in ["<0", "<1us", "<2us", "<5us", "<20us", "<100us", ">=100us"]:
    n = sum(1 for g, *_ in gaps if (k == "<0" and g < 0) or (k == "<1us" and 0 <= g < 1000) or (k == "<2us" and 1000 <= g < 2000) or (k == "<5us" and 2000 <= g < 5000) or (k == "<20us" and 5000 <= g < 20000) or (k == "<100us" and 20000 <= g < 100000) or (k == ">=100us" and g >= 100000))
    print(f"  gaps {k:8s} n={n:6d} total {h[k]/1e6:8.3f} ms")
for g, a, b, at in sorted(gaps, key=lambda x: -x[0])[:14]:
    print(f"  {g/1e3:9.1f} us at +{at/1e6:8.2f} ms  after {a}  before {b}")
