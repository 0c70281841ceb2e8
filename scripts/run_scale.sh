#!/bin/bash
# The 1/2/4/8-GPU curve of BASELINE.json in one command (run on an 8-GPU MI355X node):
#   scripts/run_scale.sh [out-dir]
# For N in 1 2 4 8: the metric's line — STRONG scaling of the 99,997,497-row ClickBench table (bench.py's default for
# N > 1: contiguous row-range shards of one table) with both exchange steps (COUNT(*) all-reduce, hit-mask all-gather) through
# torch.distributed and through the library's own C ABI (lc_comm_*); the weak-scaling curve (every rank scans its own
# 99,997,497-row column) as the secondary; and the strong split of BASELINE config 4 (TPC-H Q6 shape, 600,037,902 rows).
# One JSON line per run.
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-$R/gpurun_out/scale}
mkdir -p "$O"
: > "$O/scale_lines.jsonl"
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29500
run() {  # n, tag, bench args...
  local n=$1 tag=$2; shift 2
  port=$((port + 1))
  if [ "$n" = 1 ]; then
    python "$R/bench.py" --gpus 1 "$@" > "$O/${tag}_n1.json" 2> "$O/${tag}_n1.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $port \
      "$R/bench.py" --gpus "$n" "$@" > "$O/${tag}_n$n.json" 2> "$O/${tag}_n$n.err"
  fi
  # the compact record of the run (bench.py's last stdout line, the one a driver parses) -> scale_lines.jsonl, one line per run
  python - "$O/${tag}_n$n.json" "$tag" "$n" "$O/scale_lines.jsonl" <<'PY'
import json, sys
try:
    line = open(sys.argv[1]).read().strip().splitlines()[-1]
    d = json.loads(line)
    d["run"] = sys.argv[2]
    open(sys.argv[4], "a").write(json.dumps(d) + "\n")
    print("%-26s n=%s value %.4g %s  ms/step %.4f  exchange_by %s" % (sys.argv[2], sys.argv[3], d["value"], d["unit"], d["ms_per_step"],
                                                                      str(d["config"].get("exchange_by"))[:60]))
except Exception as e:  # noqa: BLE001
    print(sys.argv[2], "n=" + sys.argv[3], "FAILED", e)
PY
}
for n in 1 2 4 8; do
  run $n url_like_default --steps 20 --warmup 5 --no-secondary --no-cpu-baseline   # what the driver runs: strong split, --comm abi
  for comm in torch abi; do
    for ex in count mask; do
      run $n "url_like_strong_${ex}_${comm}" --steps 40 --warmup 8 --no-secondary --no-cpu-baseline --exchange $ex --comm $comm
    done
  done
  run $n url_like_weak_count_torch --scaling weak --steps 40 --warmup 8 --no-secondary --no-cpu-baseline
  run $n tpch_q6_strong --workload tpch_q6 --steps 20 --warmup 4
done
