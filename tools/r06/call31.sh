#!/bin/bash
# round 6, GPU call 31: `python bench.py --gpus 2` on a ONE-GPU box must fail loudly, and the self-launch at --gpus 1 under the driver's launch line still works
O=gpurun_out/r06c31; mkdir -p $O
timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-tail > $O/gpus2.out 2> $O/gpus2.err; echo "rc=$?" >> $O/gpus2.err
echo "--- stdout:"; cat $O/gpus2.out; echo "--- stderr tail:"; tail -4 $O/gpus2.err
