#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { name=$1; shift; echo "== $name"; "$@" > gpurun_out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 3 gpurun_out/$name.log | cut -c1-700; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
[ -n "$SKIP_PYTEST" ] || step pytest_gpu timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600
# rehearsal of the driver's multi-GPU launch line with one rank: torchrun + gloo bootstrap + RCCL communicator
export MISLAM_BENCH_FORCE_DIST=1
step bench_dist1_auto timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2
step bench_dist1_target_brute timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 3 --warmup 1 --nn brute --shard target --points 200000
step bench_dist1_target_tree timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 5 --warmup 2 --nn tree --shard target
# the same launch line with 2 and 4 ranks, all on device 0 over the gloo exchange context (RCCL refuses two ranks on one device):
# sharding, barriers, max-over-ranks timing and the report of the N > 1 path; the numbers are not measurements
unset MISLAM_BENCH_FORCE_DIST
[ -n "$ONE_RANK_ONLY" ] && exit 0
export MISLAM_BENCH_DEVICE=0 MISLAM_BENCH_TRANSPORT=gloo MISLAM_BENCH_CPD=1
step bench_rehearsal2_auto timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29536 bench.py --gpus 2 --steps 10 --warmup 2
step bench_rehearsal4_auto timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29537 bench.py --gpus 4 --steps 10 --warmup 2
# ... and WITHOUT a launcher: bench.py starts its own ranks (what an unattended N-GPU lease may run)
step bench_selflaunch2 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-whole-call
step bench_rehearsal2_target_brute timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29538 bench.py --gpus 2 --steps 3 --warmup 1 --nn brute --shard target --points 200000
exit 0
