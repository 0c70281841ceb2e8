#!/bin/bash
# Dry run of bench.py's multi-rank path on ONE GPU (ranks share device 0, gloo collectives): launcher, strong-scaling split,
# exchange and reporting logic.  Not a measurement.  usage: scripts/scale_dryrun.sh [N] [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-2}; shift
cd $R
LC_BENCH_TEST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus $N --steps 5 --warmup 2 --no-cold "$@"
