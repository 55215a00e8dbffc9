#!/bin/bash
# Build the host SIMT-emulation of the kernel sources (test infrastructure only):
# the same .hip files as libfeta_hip.so, compiled as C++ against tools/simt/hip/hip_runtime.h
# and tools/simt/feta_device.h.  Output: tools/simt/libfeta_emu.so (git-ignored).
set -e
cd "$(dirname "$0")/../.."
SRC=$(ls feta_tmlr_amd/csrc/*.hip)
g++ -O1 -g -std=c++17 -fPIC -shared -Wall -Wno-unused-variable -Wno-unknown-pragmas \
    -Itools/simt -Iinclude -Ifeta_tmlr_amd/csrc \
    -x c++ $SRC tools/simt/simt_runtime.cpp -o tools/simt/libfeta_emu.so
echo "built tools/simt/libfeta_emu.so"
