#!/bin/bash
# builds the section-profiling variant of the library (cycle stamps per solver section)
set -e
cd "$(dirname "$0")/.."
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DHOQP_PROFILE -Iinclude \
  -mllvm -sink-insts-to-avoid-spills -mllvm -disable-machine-licm \
  -o quadruped_locomotion_amd/libhoqp_hip_prof.so quadruped_locomotion_amd/csrc/hoqp_kernels.hip
