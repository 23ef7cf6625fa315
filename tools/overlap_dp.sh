#!/bin/bash
# gradient exchange on one GPU (RCCL, world size 1, collectives forced): step times for several launch thresholds / wire formats / CU reservations
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/r03_overlap_dp.json
for mb in 0 64 192 1024; do
  echo "# min_launch_mb=$mb" >> gpurun_out/r03_overlap_dp.json
  timeout 600 python tools/overlap_dp.py --steps 6 --min-launch-mb $mb --modes none,all_reduce,all_reduce_bf16 >> gpurun_out/r03_overlap_dp.json 2> gpurun_out/overlap_dp.err
done
echo "# reserved_cus=8, min_launch_mb=192" >> gpurun_out/r03_overlap_dp.json
timeout 300 python tools/overlap_dp.py --steps 6 --reserved-cus 8 --min-launch-mb 192 --modes none,all_reduce >> gpurun_out/r03_overlap_dp.json 2>> gpurun_out/overlap_dp.err
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_overlap_dp.json
