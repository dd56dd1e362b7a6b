#!/bin/bash
# Refresh the judged profiles for the bench workload: kernel-trace stats, HBM PMC passes, plain bench line.
#   bash tools/profile_round.sh r01c        (on the GPU box, from the repo root)
tag=${1:-rXX}
set -e
cd /tmp && export TMPDIR=/tmp
# the traced runs launch eagerly (--no-graph: per-kernel attribution needs it) but must consist of the kernels of the
# replayed step: queue the Linear weight gradients as a captured step does
export FOD_WGRAD_QUEUE_EAGER=1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-graph > $out/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-graph > $out/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-graph > $out/pmc_write.log 2>&1
echo "write done"
cd $R
python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write 2 $out/pmc_traffic.json > $out/pmc_traffic.txt
cp $out/pmc_traffic.json profiles/pmc_traffic.json   # so that the bench line below carries roofline.traffic
python bench.py > $out/bench_line.json 2> $out/bench.err
# MFMA-busy per entry point (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x kernel cycles)
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-graph > $out/pmc_mfma.log 2>&1
cd $R
python tools/pmc_mfma.py $out/pmc_mfma 2 > $out/pmc_mfma_busy.txt
rm -rf $out/pmc_mfma
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/stats/*/*kernel_trace.csv $out/pmc_fetch $out/pmc_write
echo "bench done"
