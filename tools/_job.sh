set -e
o=gpurun_out/r03l4
mkdir -p $o
timeout -k 10 300 python tools/bench_ops.py conv > $o/def.txt 2>&1
FOD_NT_BIG256=2 timeout -k 10 300 python tools/bench_ops.py conv > $o/sq_all.txt 2>&1
paste <(grep -E "^layer4" $o/def.txt | awk '{print $1, $2, $9, $10}') <(grep -E "^layer4" $o/sq_all.txt | awk '{print $9, $10}')
