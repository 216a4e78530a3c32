#!/bin/bash
# round 6, GPU call 23: hot vs cold operands per launch
O=gpurun_out/r06c23; mkdir -p $O
timeout 900 python tools/r06/cold_weights.py > $O/cold_weights.txt 2>$O/err.txt
cat $O/cold_weights.txt; tail -5 $O/err.txt
