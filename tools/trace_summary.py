"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid): count, total ms, median/min us."""
import collections, csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else glob.glob('gpurun_out/prof_trace/*/*kernel_trace.csv')[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    for a, b in (("_ZN12_GLOBAL__N_1", ""), ("(anonymous namespace)::", ""), ("void ", "")):
        n = n.replace(a, b)
    key = (n[:58], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {len(rows)} dispatches")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    v2 = sorted(v)
    print(f"{k[0]:58s} grid={str(k[1:]):18s} n={len(v):5d} total={sum(v)/1e3:8.3f} ms  med={v2[len(v2)//2]:8.1f} us min={v2[0]:7.1f}")
