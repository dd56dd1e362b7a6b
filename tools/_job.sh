set -e
mkdir -p gpurun_out/r03e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fp8 or two_threads" > gpurun_out/r03e/fp8_tests.log 2>&1 || { tail -40 gpurun_out/r03e/fp8_tests.log; exit 1; }
tail -2 gpurun_out/r03e/fp8_tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03e/bench_headline.json 2> gpurun_out/r03e/bench_headline.err || { tail -30 gpurun_out/r03e/bench_headline.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03e/bench_headline.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "kernels_per_replayed_step", "device_ms_per_step_profiled")})
print("roofline", d.get("roofline"), d.get("roofline_invalid"))
for k, v in d["kernel_breakdown"].items():
    print("  ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
timeout -k 10 600 python tools/ddp_overlap_probe.py 3 10 > gpurun_out/r03e/ddp_overlap_probe.txt 2>&1 || { tail -30 gpurun_out/r03e/ddp_overlap_probe.txt; exit 1; }
grep "ms/step" gpurun_out/r03e/ddp_overlap_probe.txt
timeout -k 10 900 python tools/divergence_control.py 6 200 > gpurun_out/r03e/divergence_control.txt 2>&1 || { tail -40 gpurun_out/r03e/divergence_control.txt; exit 1; }
grep -v "amdgpu.ids\|Warning\|run_backward" gpurun_out/r03e/divergence_control.txt | tail -50
