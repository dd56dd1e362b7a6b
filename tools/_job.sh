set -e
o=gpurun_out/r03tnq
mkdir -p $o
FOD_TN_BIG_SPLITS=14 FOD_TN_BIG256=0 timeout -k 10 300 python tools/bench_ops.py conv > $o/rect14.txt 2>&1
FOD_TN_BIG_SPLITS=14 FOD_TN_BIG256=1 timeout -k 10 300 python tools/bench_ops.py conv > $o/sq14.txt 2>&1
paste <(grep -E "^layer3.1.conv2|^layer4.1.conv2|^layer3.0.conv2" $o/rect14.txt | awk '{print $1, $11, $12}') <(grep -E "^layer3.1.conv2|^layer4.1.conv2|^layer3.0.conv2" $o/sq14.txt | awk '{print $11, $12}')
