#!/bin/bash
# Builds walker variants into build/variants/<name>.so: tools/variants.sh name "-DWALK_LEAF_BATCH=3 -DWALK_WAVES_PER_SIMD=5" ...
set -e
cd "$(dirname "$0")/../ray-tracer_amd/csrc"
mkdir -p ../../build/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize $flags -Rpass-analysis=kernel-resource-usage -c trace_kernels.hip -o ../../build/variants/$name.o 2> ../../build/variants/$name.txt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/$name.so api.o switches.o scene_build.o ply_reader.o analytic_kernels.o gpu_build.o ../../build/variants/$name.o
  echo "$name: $(grep -A8 'walk_kernelILi0ELi3ELi0' ../../build/variants/$name.txt | grep -E 'VGPRs:|Scratch|Occupancy' | sed 's/.*remark: *//; s/ \[.*//' | tr '\n' ' ')"
done
