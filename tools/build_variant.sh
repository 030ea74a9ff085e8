#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>  -> pathtracer_amd/lib/variants/libptgpu_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../pathtracer_amd/csrc"
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -I../../include -I. -Wno-unused-function "$@" -x hip -shared -o ../lib/variants/libptgpu_$name.so pt_kernels.hip ptgpu_api.hip pt_bvh.cc pt_scene_build.cc pt_lbvh.hip
