set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-r02x_trace}
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $out/log.txt 2>&1
cp $(ls $out/kt/*/*kernel_trace.csv | head -1) $out/kernel_trace.csv
rm -rf $out/kt
cd $R
python tools/trace_gaps.py $out/kernel_trace.csv 0.9 > $out/gaps.txt
tail -3 $out/log.txt
cat $out/gaps.txt
