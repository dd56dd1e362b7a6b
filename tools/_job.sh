mkdir -p gpurun_out/r02xd
for i in 1 2 3 4 5 6 7 8; do FOD_WGRAD_QUEUE=0 SOAK_PROBE=1 SOAK_LR=1e-5 timeout -k 10 200 python tools/soak_graph.py 200 > gpurun_out/r02xd/p_$i.txt 2>&1; echo "lr 1e-5 run $i rc=$? $(grep -E 'non-finite losses|soak ok' gpurun_out/r02xd/p_$i.txt | cut -c1-60)"; done
true
