set -e
mkdir -p gpurun_out/r03sc1
timeout -k 10 240 tools/bin/probe_sc1_16b > gpurun_out/r03sc1/probe.txt 2>&1
cat gpurun_out/r03sc1/probe.txt
