set -e
o=gpurun_out/r03perm
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_graph_gpu.py -m gpu -x -q > $o/test.log 2>&1 || { tail -40 $o/test.log; exit 1; }
tail -2 $o/test.log
for v in 4 2 1 4 2 1; do
  FOD_LN_BWD_GROUPS=$v python bench.py --no-cpu-baseline --no-extras --no-roofline 2> $o/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FOD_LN_BWD_GROUPS=$v', round(d['value'],2), round(d['ms_per_step'],3))" | tee -a $o/ab.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $GRAFT_REPO_ROOT/$o/kt.log 2>&1
cd $GRAFT_REPO_ROOT
grep -h -E "multi_permute3|ln_bwd|mlp2|multi_adamw" $(ls $o/kt/*/*kernel_stats.csv | head -1) | cut -c1-200 > $o/kstats.txt
rm -rf $o/kt
cat $o/kstats.txt
