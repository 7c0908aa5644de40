#!/bin/bash
# Development build of the library with extra compiler flags, for A/B timing on the GPU box (PTAMD_LIB=path-tracer_amd/dev/<name>.so):
#   tools/build_variant.sh <name> [flags, e.g. -DPT_STEAL_MIN=4]
# The host sources are compiled once into /tmp/ptamd_variant_obj; only the kernel file is recompiled per variant.
set -e
name=$1; shift
cd "$(dirname "$0")/../path-tracer_amd"
obj=/tmp/ptamd_variant_obj
mkdir -p $obj dev
FLAGS="-std=c++17 -O3 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -falign-loops=64"
for f in pt_host bvh_build gltf_loader jpeg_decode; do
  if [ ! -f $obj/$f.o ] || [ csrc/$f.cpp -nt $obj/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c csrc/$f.cpp -o $obj/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "$@" -c csrc/pt_kernels.hip -o $obj/k_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $obj/k_$name.o $obj/pt_host.o $obj/bvh_build.o $obj/gltf_loader.o $obj/jpeg_decode.o -lz -lpthread -o dev/$name.so
echo "built path-tracer_amd/dev/$name.so"
