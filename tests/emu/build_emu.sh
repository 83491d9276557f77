#!/bin/sh
# TEST INFRASTRUCTURE: compile the unmodified kernel source against the SIMT emulator
# (tests/emu/hip/hip_runtime.h) with g++ (-O1, no debug info: half the build time of -O2 -g, same run time).  Output: tests/emu/libjss_emu.so, same C ABI,
# "device" pointers are host pointers.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
g++ -x c++ -std=c++17 -O1 -fPIC -shared -Wall -Wno-unused-function -Wno-unknown-pragmas \
    -I"$here" -I"$root/include" "$root/jssenv_amd/csrc/jss_kernels.hip" -o "$here/libjss_emu.so"
