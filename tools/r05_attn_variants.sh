#!/bin/bash
# Lab builds of attention.hip (same ABI, separate .so under tools/bin/), selected at run time with FOURM_HIP_LIB.
#   round 5, first set:  s1: sched_barrier between q-blocks   ilp: -mllvm -amdgpu-sched-strategy=max-ilp   mem: max-memory-clause
#   round 5, last set:   nodot2: delta = rowsum(dO o O) without v_dot2c_f32_bf16
# usage: tools/r05_attn_variants.sh name "flags" [name "flags" ...]
cd "$(dirname "$0")/.."
R=$(pwd); B=$R/ml-4m_amd/build; mkdir -p tools/bin
OBJS=$(ls $B/*.o | grep -v attention)
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -x hip -I $R/include -I $R/ml-4m_amd/csrc"
build() { # name, extra flags
  /opt/rocm/bin/hipcc $CF $2 -c ml-4m_amd/csrc/attention.hip -o tools/bin/attention_$1.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libfourm_hip_$1.so $OBJS tools/bin/attention_$1.o && echo built $1
}
if [ $# -eq 0 ]; then set -- s1 "-DATTN_V2_SCHED=1" ilp "-mllvm -amdgpu-sched-strategy=max-ilp" mem "-mllvm -amdgpu-sched-strategy=max-memory-clause"; fi
while [ $# -ge 2 ]; do build "$1" "$2" & shift 2; done
wait
