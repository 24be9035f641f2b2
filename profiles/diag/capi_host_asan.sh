#!/bin/bash
# Host-side AddressSanitizer build of the library plus the C++ driver that calls every entry point
# (device code is not instrumented).  Build here:  bash profiles/diag/capi_host_asan.sh build
# Run on the GPU box:                              bash profiles/diag/capi_host_asan.sh run
set -e
cd "$(dirname "$0")/../.."
OUT=profiles/diag/asan_bin
SAN="-Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -g"
if [ "$1" = build ]; then
  mkdir -p $OUT
  make -s -C marl_amd/csrc -j4 HIPCC="/opt/rocm/bin/hipcc $SAN" OBJDIR=../../build/asan_obj OUT=../../$OUT/libmarl_amd_asan.so
  /opt/rocm/bin/hipcc $SAN -O1 -std=c++17 --offload-arch=gfx950 tests/native/capi_driver.cpp -o $OUT/capi_host_asan \
    -L$OUT -lmarl_amd_asan -Wl,-rpath,'$ORIGIN'
  ls -la $OUT
else
  # leaks are not looked for (the HIP runtime keeps its own allocations until exit)
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 timeout -k 10 300 $OUT/capi_host_asan
fi
