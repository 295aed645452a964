#!/bin/bash
out=gpurun_out/r02p; mkdir -p $out; rm -f $out/*
echo "== forced slab leg, one rank, strong 513" > $out/slab.txt
HJ_BENCH_FORCE_SLAB=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --repeats 3 >> $out/slab.txt 2> $out/slab1.err; echo "rc=$?" >> $out/slab.txt
echo "== --global-n 201 (strong, one rank)" >> $out/slab.txt
timeout -k 10 300 python bench.py --global-n 201 --steps 10 --warmup 2 --repeats 3 >> $out/slab.txt 2> $out/slab2.err; echo "rc=$?" >> $out/slab.txt
echo "== weak leg, one rank" >> $out/slab.txt
HJ_BENCH_FORCE_SLAB=1 timeout -k 10 300 python bench.py --global-n 0 --steps 10 --warmup 2 --repeats 3 >> $out/slab.txt 2> $out/slab3.err; echo "rc=$?" >> $out/slab.txt
echo "== torchrun 1 rank" >> $out/slab.txt
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 >> $out/slab.txt 2> $out/slab4.err; echo "rc=$?" >> $out/slab.txt
cut -c1-700 $out/slab.txt
tail -3 $out/slab1.err $out/slab2.err $out/slab3.err
