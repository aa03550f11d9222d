for lk in 1 0; do
VCA_LINEAR=$lk CONV_BENCH_ROWS=400 timeout -k 10 500 python -u tools/conv_bench.py mixed > gpurun_out/r3_conv_layers_graph_lin$lk.txt 2>&1
done
