#!/bin/bash
# Lab builds of attention.hip (same ABI, separate .so under tools/bin/): scheduling variants of the 128 x 128 backward.
#   s1: sched_barrier between q-blocks      ilp: -mllvm -amdgpu-sched-strategy=max-ilp      mem: max-memory-clause
cd "$(dirname "$0")/.."
R=$(pwd); B=$R/ml-4m_amd/build; mkdir -p tools/bin
OBJS=$(ls $B/*.o | grep -v attention)
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -x hip -I $R/include -I $R/ml-4m_amd/csrc"
build() { # name, extra flags
  /opt/rocm/bin/hipcc $CF $2 -c ml-4m_amd/csrc/attention.hip -o tools/bin/attention_$1.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libfourm_hip_$1.so $OBJS tools/bin/attention_$1.o && echo built $1
}
build s1 "-DATTN_V2_SCHED=1" &
build ilp "-mllvm -amdgpu-sched-strategy=max-ilp" &
build mem "-mllvm -amdgpu-sched-strategy=max-memory-clause" &
wait
