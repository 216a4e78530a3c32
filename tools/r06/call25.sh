#!/bin/bash
# round 6, GPU call 25: the whole GPU suite once more on the final tree (library digest unchanged since call 17; + tests/test_gpu_library_path.py), then smoke()
O=gpurun_out/r06c25; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu --durations=10 > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
