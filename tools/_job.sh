mkdir -p gpurun_out/r02xw
for w in cfg2 nusc500-stage1 nusc500-stage2 t8; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r02xw/bench_$w.json 2> gpurun_out/r02xw/bench_$w.err || tail -5 gpurun_out/r02xw/bench_$w.err
done
timeout -k 10 300 python bench.py --train-mode --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r02xw/bench_trainmode_headline.json 2> gpurun_out/r02xw/bench_trainmode_headline.err
timeout -k 10 300 python bench.py --workload cfg2 --train-mode --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r02xw/bench_trainmode_cfg2.json 2> gpurun_out/r02xw/bench_trainmode_cfg2.err
timeout -k 10 300 python bench.py --workload nusc500-stage1 --train-mode --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r02xw/bench_trainmode_nusc500-stage1.json 2> gpurun_out/r02xw/bench_trainmode_nusc500-stage1.err
timeout -k 10 300 python bench.py --forward-only --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r02xw/bench_forward_headline.json 2> gpurun_out/r02xw/bench_forward_headline.err
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02xw/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],2))
    except Exception as e: print(f, "ERR", e)
P
true
