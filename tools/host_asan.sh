#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the library (GPU code is not instrumented): builds libptamd.so with the
# sanitizers on the host pass only and the headless CLI against it, into path-tracer_amd/dev/ (CPU container, repo root):
#   tools/host_asan.sh build
# then, on the GPU box, renders a few scene files through commit (tree build, eight-child collapse, uploads), render,
# film read-back and PNG output:
#   tools/host_asan.sh run scene1.txt scene2.txt ...
set -e
D=path-tracer_amd/dev
if [ "$1" = build ]; then
  mkdir -p $D
  (cd path-tracer_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O1 -g -fPIC -Wall -Wno-unused-function -fno-slp-vectorize \
     -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -shared csrc/pt_kernels.hip csrc/pt_host.cpp csrc/bvh_build.cpp \
     csrc/gltf_loader.cpp csrc/jpeg_decode.cpp -lz -lpthread -o dev/libptamd.so
   /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-gpu-sanitize host/ptamd_cli.cpp -o dev/ptamd_cli_asan -I. -Ldev -lptamd -lz -Wl,-rpath,'$ORIGIN')
  exit 0
fi
shift
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
for s in "$@"; do
  echo "== $s"
  timeout -k 10 120 $D/ptamd_cli_asan "$s" --out /tmp/host_asan.png --spp 4 --depth 6 --frames
done
