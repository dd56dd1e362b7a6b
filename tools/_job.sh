set -e
o=gpurun_out/r03qb
mkdir -p $o
F="--no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 3"
for w in cfg2 nusc500-stage1 nusc500-stage2; do
  timeout -k 10 300 python bench.py --workload $w $F > $o/bench_$w.json 2> $o/err_$w.txt
  python -c "import json,sys; d=json.loads(open('$o/bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(d['value'],1), round(d['ms_per_step'],2))"
done
timeout -k 10 300 python bench.py --workload t8 --attn-dtype fp8 $F > $o/bench_t8_fp8.json 2> $o/err_t8.txt
python -c "import json,sys; d=json.loads(open('$o/bench_t8_fp8.json').read().strip().splitlines()[-1]); print('t8 fp8', round(d['value'],1), round(d['ms_per_step'],2), d.get('vs_bf16_attention'))"
timeout -k 10 300 python bench.py --train-mode $F > $o/bench_trainmode_headline.json 2> $o/err_tm.txt
python -c "import json,sys; d=json.loads(open('$o/bench_trainmode_headline.json').read().strip().splitlines()[-1]); print('train', round(d['value'],1), round(d['ms_per_step'],2))"
timeout -k 10 300 python bench.py --force-ddp $F > $o/bench_force_ddp.json 2> $o/err_ddp.txt
python -c "import json,sys; d=json.loads(open('$o/bench_force_ddp.json').read().strip().splitlines()[-1]); print('force-ddp', round(d['value'],1), round(d['ms_per_step'],2))"
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "library_imported" 2>&1 | tail -1
python __graft_entry__.py smoke 2>&1 | tail -1
