#!/bin/bash
# Rehearsal of `bench.py --gpus 3` on ONE GPU (every rank on device 0, gloo for the all-gather), with two batches in
# flight and with --no-pipeline; writes gpurun_out/reh3*.json.  Run on the GPU box.
set -e
export SQUISH_DEVICE=0 SQUISH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 3 --steps 6 --warmup 2 --chunks 20000 > gpurun_out/reh3.json 2> gpurun_out/reh3.err
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 3 --steps 6 --warmup 2 --chunks 20000 --no-pipeline > gpurun_out/reh3_nopipe.json 2> gpurun_out/reh3_nopipe.err
