#!/bin/bash
# Multi-GPU bring-up kit: nothing of the N > 1 path has met a second device yet.  Run on a node with N GPUs
# (default 8), from the repository root, after `python -c 'import __graft_entry__ as g; g.build()'`.  Stops at the first failure; a
# failing rank has printed its sarpro_hip_last_error.  Order: the cheapest thing that can fail first.
#   1  communicator self-test over RCCL, 2 ranks, then N                  (tools/bringup_rank.py comm)
#   2  403 x 520 scene as row stripes over real RCCL against the oracle     (tools/bringup_rank.py stripes), 2 ranks, then N
#   3  the in-process batch driver on two DIFFERENT devices                 (tests/test_gpu_multi_worker.py, needs >= 2 GPUs)
#   4  bench.py --gpus 2 / 4 / N, batch mode (weak scaling) and stripe mode (strong scaling); the JSON lines are kept
# usage: tools/bringup_8gpu.sh [N] [--dry-run]     --dry-run: the launch commands only (what the CPU suite checks)
set -u
N=${1:-8}; DRY=0; [ "${2:-}" = "--dry-run" ] && DRY=1
cd "$(dirname "$0")/.."
OUT=gpurun_out/bringup; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29540
run() { # run <name> <command...>
  local name=$1; shift
  echo "== $name: $*"
  [ $DRY = 1 ] && return 0
  "$@" > $OUT/$name.log 2>&1
  local rc=$?
  if [ $rc != 0 ]; then echo "FAILED ($rc): $name -- last lines:"; tail -20 $OUT/$name.log; exit $rc; fi
  tail -3 $OUT/$name.log
}
launch() { echo python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $PORT; }
for n in 2 $N; do PORT=$((PORT + 1)); run comm_$n $(launch $n) tools/bringup_rank.py comm; done
for n in 2 $N; do PORT=$((PORT + 1)); run stripes_$n $(launch $n) tools/bringup_rank.py stripes; done
run multi_worker python -m pytest tests/test_gpu_multi_worker.py -x -q -m gpu
for n in 2 4 $N; do
  [ $n -gt $N ] && continue
  run bench_batch_$n python bench.py --gpus $n --steps 10 --warmup 3 --mode batch
  run bench_stripe_$n python bench.py --gpus $n --steps 10 --warmup 3 --mode stripe
done
[ $DRY = 1 ] || { echo "bring-up passed on $N GPUs; bench lines:"; grep -h '^{' $OUT/bench_*.log | cut -c1-400; }
