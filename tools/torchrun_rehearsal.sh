cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out; mkdir -p $O
export LADI_FORCE_COLLECTIVE=1
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r04_torchrun_rehearsal_config1.json 2> $O/rehearsal1.err
tail -c 600 $O/r04_torchrun_rehearsal_config1.json; echo; tail -3 $O/rehearsal1.err
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --config 3 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-tail > $O/r04_torchrun_rehearsal_config3.json 2> $O/rehearsal3.err
head -c 400 $O/r04_torchrun_rehearsal_config3.json; echo; tail -3 $O/rehearsal3.err
