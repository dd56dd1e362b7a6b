#!/bin/bash
# SQ counters of the attention kernels on the encoder shape: where the wave cycles go (issue / wait / stall).
# usage (on the GPU box): bash tools/attn_pmc.sh <tag> [attn | attn8]      (attn8: the fp8 forward beside the bf16 one)
set -e
tag=${1:-attn}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $out/pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py ${2:-attn} > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out/pmc2 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py ${2:-attn} > $out/pmc2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("pmc1", "pmc2"):
    files = glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "attn" not in k: continue
            import re
            m_ = re.search(r"(attn_\w+<[^>]*>)", k)
            key = (m_.group(1) if m_ else k[:60], r.get("Grid_Size"))
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    for key, c in agg.items():
        n = None
        print(d, key)
        for name, v in sorted(c.items()):
            print(f"    {name:28s} {v:16.0f}")
PY
