set -e
o=gpurun_out/r03zb
mkdir -p $o
python bench.py --workload t8 --attn-dtype fp8 --no-cpu-baseline --no-extras --no-roofline > $o/bench_t8_fp8.json 2> $o/bench_t8_fp8.err
for w in cfg2 nusc500-stage1 nusc500-stage2; do
  python bench.py --workload $w --no-cpu-baseline --no-extras --no-roofline > $o/bench_$w.json 2> $o/bench_$w.err
done
python bench.py --train-mode --no-cpu-baseline --no-extras --no-roofline > $o/bench_trainmode_headline.json 2> $o/bench_trainmode_headline.err
python bench.py --force-ddp --no-cpu-baseline --no-extras --no-roofline > $o/bench_force_ddp.json 2> $o/bench_force_ddp.err
for f in $o/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d['value'],2), round(d['ms_per_step'],3), (d.get('vs_bf16_attention') or {}).get('throughput_ratio_fp8_over_bf16'))
PY
done
