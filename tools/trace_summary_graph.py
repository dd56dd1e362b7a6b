"""Per-kernel totals of ONE replayed step from a rocprofv3 kernel trace of `bench.py` in graph mode (tools/trace_graph.sh):
the last step = the kernels between the last two AdamW launches."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "multi_adamw" in r["Kernel_Name"]]
seg = rows[ad[-2] + 1:ad[-1] + 1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"void ", "", n); n = re.sub(r"at::native::", "", n)
    return n[:86]


agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = short(r["Kernel_Name"]); agg[k][0] += d; agg[k][1] += 1
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
busy = sum(v[0] for v in agg.values()) / 1e6
print(f"one replayed step under rocprofv3 --kernel-trace: {len(seg)} kernels, span {span:.2f} ms, kernel time {busy:.2f} ms")
print(f"{'ms/step':>8s} {'launches':>8s} {'avg us':>8s}  kernel")
for k, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
    print(f"{d / 1e6:8.3f} {c:8d} {d / c / 1e3:8.1f}  {k}")

# the same step per C-ABI entry point, with bench.py's own kernel -> entry mapping: these are the figures bench.py's
# `roofline` / `kernel_breakdown` must reproduce (within 5 %)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _entry_of  # noqa: E402

ent = collections.defaultdict(lambda: [0, 0])
for r in seg:
    e = _entry_of(r["Kernel_Name"])
    ent[e][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); ent[e][1] += 1
print()
print(f"{'ms/step':>8s} {'launches':>8s} {'avg us':>8s}  C-ABI entry point (bench.py mapping)")
for k, (d, c) in sorted(ent.items(), key=lambda x: -x[1][0]):
    print(f"{d / 1e6:8.3f} {c:8d} {d / c / 1e3:8.1f}  {k}")
