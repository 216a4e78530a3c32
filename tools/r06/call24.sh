#!/bin/bash
# round 6, GPU call 24: LDS-DMA staging vs register staging under MFMA load (tools/r06/stage_path.hip)
O=gpurun_out/r06c24; mkdir -p $O
timeout 300 tools/r06/bin/stage_path > $O/stage_path.txt 2>&1
cat $O/stage_path.txt
