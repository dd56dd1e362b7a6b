#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in tn tnconv nt; do
  rm -rf gpurun_out/pmc_$w
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_$w -- python tools/pmc_one.py $w > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc2_$w -- python tools/pmc_one.py $w > /dev/null 2>&1
  python - "$w" <<'PY'
import csv, glob, sys, collections
w = sys.argv[1]
for d in (f'gpurun_out/pmc_{w}', f'gpurun_out/pmc2_{w}'):
    fs = glob.glob(d + '/*/*counter_collection.csv')
    if not fs: print(d, 'no data'); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        if 'gemm' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(w, {k: round(sum(v)/len(v)/1e6, 3) for k, v in agg.items()}, '(millions per launch)')
PY
done
