#!/bin/bash
# wgrad traffic/time experiment: time table + FETCH_SIZE total for the conv TN kernel under different knobs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ONLY_WGRAD=1
for cfg in "0 0" "1 0" "1 512" "1 1024" "0 1024"; do
  set -- $cfg
  export FOD_TN_XCD=$1 FOD_TN_ROWS=$2
  echo "=== XCD=$1 ROWS=$2"
  python tools/bench_ops.py conv 2>/dev/null | grep -E "totals|layer2.1.conv2|layer3.1.conv2|layer4.1.conv2|layer3.1.conv1|layer2.0.conv3"
  rm -rf gpurun_out/pmc_w; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python tools/bench_ops.py conv > /dev/null 2>&1
  python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/pmc_w/*/*counter_collection.csv')[0])))
tot = sum(float(r['Counter_Value']) for r in rows if r['Counter_Name']=='FETCH_SIZE' and 'gemm_tn' in r['Kernel_Name'])
n = sum(1 for r in rows if r['Counter_Name']=='FETCH_SIZE' and 'gemm_tn' in r['Kernel_Name'])
print(f"  conv wgrad FETCH (x2 corrected) total {2*tot*1024/1e9:.2f} GB over {n} launches")
PY
done
