set -e
o=gpurun_out/r03w
mkdir -p $o
F="--no-cpu-baseline --no-extras --no-roofline --steps 10 --warmup 2"
timeout -k 10 300 python bench.py --workload t8 --attn-dtype fp8 $F > $o/bench_t8_fp8.json 2> $o/err_t8.txt
python -c "import json,sys; d=json.loads(open('$o/bench_t8_fp8.json').read().strip().splitlines()[-1]); v=d.get('vs_bf16_attention'); print('t8 fp8', round(d['value'],1), round(d['ms_per_step'],2), round(v['bf16']['value'],1), round(v['bf16']['ms_per_step'],2), round(v['throughput_ratio_fp8_over_bf16'],4))"
timeout -k 10 300 python bench.py --train-mode $F > $o/bench_trainmode_headline.json 2> $o/err_tm.txt
python -c "import json,sys; d=json.loads(open('$o/bench_trainmode_headline.json').read().strip().splitlines()[-1]); print('train', round(d['value'],1), round(d['ms_per_step'],2))"
timeout -k 10 300 python bench.py --force-ddp $F > $o/bench_force_ddp.json 2> $o/err_ddp.txt
python -c "import json,sys; d=json.loads(open('$o/bench_force_ddp.json').read().strip().splitlines()[-1]); print('force-ddp', round(d['value'],1), round(d['ms_per_step'],2))"
