#!/bin/bash
# round 6: the whole GPU suite on the final binary with the re-measured tile table (parity record -> gpurun_out/parity_r06.json), then smoke()
O=gpurun_out/r06c17; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu --durations=10 > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
