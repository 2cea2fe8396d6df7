#!/bin/bash
# A variant build of libptmi for A/B timing (same flags as csrc/Makefile + the given knobs): tools/build_variant.sh NAME [-DKNOB=value ...]
# -> build/variants/NAME.so, loaded through PTMI_LIB=build/variants/NAME.so (tools/ab_*.sh)
set -e
cd "$(dirname "$0")/../path_tracer_amd/csrc"
name=$1; shift
mkdir -p ../../build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wno-unused-parameter -Wno-missing-field-initializers "$@" -shared -o ../../build/variants/$name.so \
  -x hip pt_kernels.hip -x hip pt_post.hip -x hip pt_api.cpp -x hip pt_scene.cpp -x hip pt_png.cpp
echo built build/variants/$name.so "$@"
