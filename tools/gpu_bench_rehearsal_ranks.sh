#!/bin/bash
# Only the multi-rank part of tools/gpu_dist_rehearsal.sh (2 and 4 ranks on device 0 over the gloo exchange context).
mkdir -p gpurun_out
step() { name=$1; shift; echo "== $name"; "$@" > gpurun_out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 3 gpurun_out/$name.log | cut -c1-900; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
export MISLAM_BENCH_DEVICE=0 MISLAM_BENCH_TRANSPORT=gloo MISLAM_BENCH_CPD=1
step bench_rehearsal2_auto timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29536 bench.py --gpus 2 --steps 10 --warmup 2
step bench_rehearsal4_auto timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29537 bench.py --gpus 4 --steps 10 --warmup 2
step bench_rehearsal2_target_brute timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29538 bench.py --gpus 2 --steps 3 --warmup 1 --nn brute --shard target --points 200000
exit 0
