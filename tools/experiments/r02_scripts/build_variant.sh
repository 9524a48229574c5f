#!/bin/bash
# Build a variant of libhpx.so next to the product library: tools/build_variant.sh <tag> VAR=VALUE ...
# -> hydra_pspec_amd/variants/libhpx_<tag>.so, selected at run time with HPX_LIB_PATH=<that file>.
# Only hpx_factor.hip depends on the HPX_* switches; the other objects are reused.
set -e
TAG=$1; shift
C=hydra_pspec_amd/csrc; V=hydra_pspec_amd/variants; mkdir -p $V
FLAGS=""
for kv in "$@"; do FLAGS="$FLAGS -D$kv"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c $C/hpx_factor.hip -o $V/hpx_factor_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libhpx_$TAG.so $V/hpx_factor_$TAG.o $C/hpx_transform.o $C/hpx_chain.o $C/hpx_extra.o $C/hpx_flat.o $C/hpx_lowrank.o $C/hpx_modes.o
echo built $V/libhpx_$TAG.so
