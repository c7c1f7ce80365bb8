#!/bin/bash
# Hop timings of the persistent sweep kernel (1 and 2 columns) and of the chained launches: gpurun_out/hops_*.txt
TAG=${1:-r04}
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scale-modes"
HIPKKT_CHAIN=0 HIPKKT_TOP_STAMPS=9 timeout -k 10 200 $B 2>&1 | grep "top stamps" > gpurun_out/${TAG}_hops_top_1col.txt
HIPKKT_CHAIN=0 HIPKKT_TOP_STAMPS=5 HIPKKT_TOP_STAMPS_NR=2 timeout -k 10 200 $B 2>&1 | grep "top stamps" > gpurun_out/${TAG}_hops_top_2col.txt
HIPKKT_TOP_STAMPS=9 timeout -k 10 200 $B 2>&1 | grep "chain stamps" > gpurun_out/${TAG}_hops_chain.txt
wc -l gpurun_out/${TAG}_hops_*.txt
