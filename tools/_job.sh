set -e
mkdir -p gpurun_out/r03f
for ch in 2 8; do
  NCCL_MAX_NCHANNELS=$ch timeout -k 10 400 python tools/ddp_overlap_probe.py 2 10 > gpurun_out/r03f/ddp_probe_ch$ch.txt 2>&1 || { tail -20 gpurun_out/r03f/ddp_probe_ch$ch.txt; exit 1; }
  echo "NCCL_MAX_NCHANNELS=$ch"; grep "ms/step" gpurun_out/r03f/ddp_probe_ch$ch.txt
done
timeout -k 10 900 python tools/divergence_control.py 6 200 > gpurun_out/r03f/divergence_control.txt 2>&1 || { tail -40 gpurun_out/r03f/divergence_control.txt; exit 1; }
grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/r03f/divergence_control.txt | tail -50
