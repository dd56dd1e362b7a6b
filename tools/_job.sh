set -e
mkdir -p gpurun_out/r03s
timeout -k 10 200 python tools/probe_write_bw.py > gpurun_out/r03s/wbw.txt 2>&1 || { tail -20 gpurun_out/r03s/wbw.txt; exit 1; }
cat gpurun_out/r03s/wbw.txt
