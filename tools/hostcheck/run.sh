#!/bin/bash
# Builds the library's host code with AddressSanitizer and UBSan (host side only; the device code is compiled as usual) into one
# executable with tools/hostcheck/driver.cpp and runs it. No GPU needed.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); O=${RT_HOSTCHECK_TMP:-/tmp}/rt_hostcheck_bin
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-result \
  -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
  $R/ray-tracer_amd/csrc/rt_kernels.hip $R/ray-tracer_amd/csrc/rt_host.cpp $R/ray-tracer_amd/csrc/rt_comm.cpp $R/tools/hostcheck/driver.cpp \
  -o $O -ldl
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 $O
