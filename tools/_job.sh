set -e
tag=r03s
o=gpurun_out/$tag
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1 || { tail -30 gpurun_out/${tag}_profile.log; exit 1; }
tail -1 gpurun_out/${tag}_profile.log
bash tools/trace_graph.sh ${tag}_trace > /dev/null 2>&1
python tools/trace_summary_graph.py gpurun_out/${tag}_trace/kernel_trace.csv > gpurun_out/${tag}_trace/summary.txt 2>&1
rm -f gpurun_out/${tag}_trace/kernel_trace.csv
head -1 gpurun_out/${tag}_trace/summary.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gpu_tests.txt 2>&1 || { tail -40 gpurun_out/${tag}_gpu_tests.txt; exit 1; }
tail -1 gpurun_out/${tag}_gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -1
F="--no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 3"
for w in cfg2 nusc500-stage1 nusc500-stage2; do
  timeout -k 10 300 python bench.py --workload $w $F > $o/bench_$w.json 2> $o/err_$w.txt
  python -c "import json,sys; d=json.loads(open('$o/bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(d['value'],1), round(d['ms_per_step'],2))"
done
timeout -k 10 300 python bench.py --workload t8 --attn-dtype fp8 $F > $o/bench_t8_fp8.json 2> $o/err_t8.txt
python -c "import json,sys; d=json.loads(open('$o/bench_t8_fp8.json').read().strip().splitlines()[-1]); v=d.get('vs_bf16_attention'); print('t8 fp8', round(d['value'],1), round(d['ms_per_step'],2), round(v['bf16']['value'],1), round(v['bf16']['ms_per_step'],2), round(v['throughput_ratio_fp8_over_bf16'],4))"
timeout -k 10 300 python bench.py --train-mode $F > $o/bench_trainmode_headline.json 2> $o/err_tm.txt
python -c "import json,sys; d=json.loads(open('$o/bench_trainmode_headline.json').read().strip().splitlines()[-1]); print('train', round(d['value'],1), round(d['ms_per_step'],2))"
timeout -k 10 300 python bench.py --force-ddp $F > $o/bench_force_ddp.json 2> $o/err_ddp.txt
python -c "import json,sys; d=json.loads(open('$o/bench_force_ddp.json').read().strip().splitlines()[-1]); print('force-ddp', round(d['value'],1), round(d['ms_per_step'],2))"
