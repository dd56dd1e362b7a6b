#!/bin/bash
# Refresh the judged profiles for the bench workload: kernel-trace stats, HBM PMC passes, plain bench line.
#   bash tools/profile_round.sh r01c        (on the GPU box, from the repo root)
tag=${1:-rXX}
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $out/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $out/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $out/pmc_write.log 2>&1
echo "write done"
cd $R
python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write 2 $out/pmc_traffic.json > $out/pmc_traffic.txt
cp $out/pmc_traffic.json profiles/pmc_traffic.json   # so that the bench line below carries roofline.traffic
python bench.py > $out/bench_line.json 2> $out/bench.err
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/stats/*/*kernel_trace.csv $out/pmc_fetch $out/pmc_write
echo "bench done"
