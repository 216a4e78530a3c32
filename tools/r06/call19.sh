#!/bin/bash
# round 6, GPU call 19: is the loop's forward slower than the forward alone because of SUSTAINED load?  bench.py --roofline-only with 8 and with 100 back-to-back forwards
# (hipGraph replay, unet_forward_lanes_ms), this tree and the round-5 library, interleaved
for rep in 1 2; do for arm in base new; do for it in 8 100; do
  if [ $arm = base ]; then D=tools/r06/base_r05; else D=.; fi
  ( cd $D && timeout 600 python bench.py --roofline-only --no-cpu-baseline --roofline-iters $it 2> /dev/null | tail -1 ) | python -c "
import json,sys
r=json.loads(sys.stdin.read())['roofline']
print('$arm', 'iters', $it, 'eager_ms', r['unet_forward_ms'], 'graph_replay_ms', r['unet_forward_lanes_ms'], 'clock_probe_mhz', r['clock'].get('under_unet_forward_mhz'))"
done; done; done
