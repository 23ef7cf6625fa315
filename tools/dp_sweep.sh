#!/bin/bash
# First thing to run on an N-GPU MI355X node (none was available in rounds 1-4): the data-parallel knobs of fourm.parallel.DataParallel against
# each other, the way the driver launches bench.py.  usage: tools/dp_sweep.sh [N=8] [steps=10]      (one JSON line per setting -> gpurun_out/dp_sweep.jsonl)
#   FOURM_DP_EXCHANGE        overlap (stage-wise exchange under the backward) | tail (one exchange after it)
#   FOURM_DP_RESERVED_CUS    CUs the persistent GEMM grids leave to RCCL in overlap mode (default 16; 0 = none)
#   NCCL_MAX_NCHANNELS       RCCL channels = resident collective workgroups (default: = reserved CUs)
#   FOURM_DP_MIN_LAUNCH_MB   gradient bytes that must be waiting before a collective is launched (default 192)
cd "$(dirname "$0")/.."
N=${1:-8}; STEPS=${2:-10}
mkdir -p gpurun_out; : > gpurun_out/dp_sweep.jsonl
run() {   # env assignments as arguments
  port=$((29500 + RANDOM % 2000))
  line=$(env "$@" timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $port \
         bench.py --gpus "$N" --steps "$STEPS" --warmup 3 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | tail -1)
  echo "{\"env\": \"$*\", \"record\": ${line:-null}}" >> gpurun_out/dp_sweep.jsonl
  echo "$* -> $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read() or "{}"); print(d.get("ms_per_step"), d.get("value"))' 2>/dev/null)"
}
run FOURM_DP_EXCHANGE=tail
run FOURM_DP_EXCHANGE=overlap FOURM_DP_RESERVED_CUS=0  NCCL_MAX_NCHANNELS=32
run FOURM_DP_EXCHANGE=overlap FOURM_DP_RESERVED_CUS=8  NCCL_MAX_NCHANNELS=8
run FOURM_DP_EXCHANGE=overlap FOURM_DP_RESERVED_CUS=16 NCCL_MAX_NCHANNELS=16
run FOURM_DP_EXCHANGE=overlap FOURM_DP_RESERVED_CUS=32 NCCL_MAX_NCHANNELS=32
run FOURM_DP_EXCHANGE=overlap FOURM_DP_RESERVED_CUS=16 NCCL_MAX_NCHANNELS=16 FOURM_DP_MIN_LAUNCH_MB=64
run FOURM_DP_EXCHANGE=overlap FOURM_DP_RESERVED_CUS=16 NCCL_MAX_NCHANNELS=16 FOURM_DP_MIN_LAUNCH_MB=512
