# kernel trace of the split-async data-parallel step at one rank: what runs beside the backbone's backward, where the gaps are
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03g
mkdir -p $out
FOD_GRAPH_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python $R/bench.py --gpus 1 --force-ddp --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > $out/trace_log.txt 2>&1
cp $(ls $out/kt/*/*kernel_trace.csv | head -1) $out/kernel_trace_overlap.csv
rm -rf $out/kt
cd $R
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03g/kernel_trace_overlap.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "nccl" in n.lower() or "rccl" in n.lower()]
print(len(rows), "kernels;", len(idx), "collective kernels")
t0 = int(rows[0]["Start_Timestamp"])
for i in idx[-8:]:
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    # kernels overlapping in time
    ov = [q for q in rows if q is not r and int(q["Start_Timestamp"]) < e and int(q["End_Timestamp"]) > s]
    print(f"collective {r['Kernel_Name'][:50]} start {(s - t0) / 1e3:.1f} us dur {(e - s) / 1e3:.1f} us grid {r.get('Grid_Size')} wg {r.get('Workgroup_Size')}; "
          f"{len(ov)} kernels overlap it; queue {r.get('Queue_Id')}")
# last step: per-kernel sum, and idle gaps > 20 us
marks = [i for i, n in enumerate(names) if "post_proc_kernel" in n]
a, b = marks[-2], marks[-1]
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
busy = 0
cur_end = int(rows[a]["Start_Timestamp"])
gaps = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end + 20000:
        gaps.append(((cur_end - int(rows[a]["Start_Timestamp"])) / 1e3, (s - cur_end) / 1e3, r["Kernel_Name"][:60]))
    cur_end = max(cur_end, e)
print(f"last step span {span / 1e6:.3f} ms; gaps > 20 us:")
for g in gaps:
    print(f"   at {g[0]:9.1f} us: {g[1]:7.1f} us idle before {g[2]}")
PY
