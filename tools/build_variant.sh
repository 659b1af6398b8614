#!/bin/bash
# tools/build_variant.sh NAME SRC.hip "FLAGS" : an alternative build of libshenqi_hip.so in build/NAME (SHQ_LIBDIR=build/NAME): SRC.hip recompiled
# with FLAGS, every other object taken from shenqi_amd/lib; libshenqi_host.so copied.  For same-box A/B runs (tools/ab_many.sh).
set -e
name=$1; src=$2; flags=$3
mkdir -p build/$name
obj=build/$name/$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Iinclude $flags -c shenqi_amd/csrc/$src -o $obj
others=$(ls shenqi_amd/lib/*.o | grep -v "/$(basename ${src%.hip}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/$name/libshenqi_hip.so $obj $others -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
cp shenqi_amd/lib/libshenqi_host.so build/$name/
echo built build/$name
