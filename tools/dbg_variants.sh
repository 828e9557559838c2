#!/bin/bash
# developer tool: build timing-only variants of the library (operands from one tile; parts of the loop removed)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_dbg
for v in BASE NOMFMA NOSTORE NOLOAD "NOLOAD -DSRN_DBG_NOSTORE"; do
  name=$(echo "$v" | tr -d ' -' | sed 's/DSRN_DBG_//')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Iinclude -Iserenade_amd/csrc \
    -DSRN_DEBUG_SAMETILE -DSRN_DBG_$v -c serenade_amd/csrc/conv_gemm.hip -o build_dbg/conv_gemm_$name.o &
done
wait
for v in BASE NOMFMA NOSTORE NOLOAD NOLOADNOSTORE; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_dbg/lib_$v.so build_dbg/conv_gemm_$v.o \
    serenade_amd/build/conv_halo.hip.o serenade_amd/build/conv_planes.hip.o serenade_amd/build/norm_act.hip.o \
    serenade_amd/build/gst.hip.o serenade_amd/build/api.cpp.o
done
ls -la build_dbg/*.so
