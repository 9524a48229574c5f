#!/bin/bash
# usage: build_variant.sh <name> <factor source .hip>   -> tools/experiments/ab/libhpx_<name>.so
# (the other objects of the product build are reused; run `make -C hydra_pspec_amd/csrc` first)
set -e
ROOT=$(cd "$(dirname "$0")/../../.." && pwd)
CS=$ROOT/hydra_pspec_amd/csrc
cp "$2" $CS/_variant_tmp.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c $CS/_variant_tmp.hip -o /tmp/_variant_$1.o
rm -f $CS/_variant_tmp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/experiments/ab/libhpx_$1.so /tmp/_variant_$1.o \
  $CS/hpx_transform.o $CS/hpx_plan.o $CS/hpx_setup.o $CS/hpx_chain.o $CS/hpx_post.o $CS/hpx_woodbury.o $CS/hpx_extra.o $CS/hpx_flat.o $CS/hpx_lowrank.o $CS/hpx_modes.o
echo built libhpx_$1.so
