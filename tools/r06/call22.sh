#!/bin/bash
# round 6, GPU call 22: vendor GEMM on the forward's plain-GEMM shapes
O=gpurun_out/r06c22; mkdir -p $O
timeout 600 python tools/r06/vendor_gemm_shapes.py > $O/vendor_gemm_shapes.txt 2>$O/err.txt
cat $O/vendor_gemm_shapes.txt; tail -3 $O/err.txt
