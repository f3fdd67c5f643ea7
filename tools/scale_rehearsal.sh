#!/bin/bash
# The 8-GPU path, one command away: no 8-GPU node has
# been available to any round of this build, so everything that depends on the
# rank count is run with 2, 4 and 8 ranks OVERSUBSCRIBED on the one GPU of the
# box -- bench.py exactly as the driver launches it, through BOTH gathers
# (torch.distributed point-to-point and the library's own communicator,
# opty_hip_gather_v, over the test transport tests/fake_rccl in librccl's
# place) and to_host --, every line must certify itself against the reference's
# golden record (config.verify.ok), and tools/scale_model.py turns the shard
# launches measured on this GPU plus the link model of DESIGN.md section 7
# into the curve the first real SCALE_r0N.json can be compared with:
# profiles/r06_scale_model.json (compute only, + gather, + to_host).
#
#   tools/scale_rehearsal.sh [out_dir]       (on a GPU box; ~6 minutes)
set -e
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/rehearsal}
mkdir -p "$OUT"
FAKE=/tmp/libfake_rccl_$$.so
hipcc --offload-arch=gfx950 -shared -fPIC -O1 tests/fake_rccl/fake_rccl.cpp -o $FAKE
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_SOCKET_IFNAME=lo
python bench.py --gpus 1 --steps 100 --warmup 10 > "$OUT/n1.json" 2> "$OUT/n1.err"
PORT=29610
for N in 2 4 8; do
  PORT=$((PORT + 7))
  OPTY_BENCH_OVERSUBSCRIBE=1 OPTY_HIP_RCCL_LIBRARY=$FAKE \
  timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N \
      --steps 10 --warmup 3 --prewarm-ms 20 --no-cpu-baseline \
      > "$OUT/n$N.json" 2> "$OUT/n$N.err" || { tail -20 "$OUT/n$N.err"; exit 1; }
  # ... and the torch-free launch of the same ranks (ctypes + side channel)
  PORT=$((PORT + 7))
  for R in $(seq 0 $((N - 1))); do
    RANK=$R LOCAL_RANK=$R WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
    OPTY_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --no-torch --gpus $N \
        --steps 10 --warmup 3 --prewarm-ms 20 \
        > "$OUT/n${N}_notorch_r$R.json" 2> "$OUT/n${N}_notorch_r$R.err" &
  done
  wait
done
rm -f $FAKE
python tools/scale_model.py "$OUT" > profiles/r06_scale_model.json
cat profiles/r06_scale_model.json | head -60
