#!/bin/bash
# Developer tool: build a timing-only / instrumented variant of the library next to the product build.
#
#   tools/dbg_variants.sh NAME SOURCE.hip -DFLAG [-DFLAG ...]
#
# compiles serenade_amd/csrc/SOURCE.hip with the flags and links it with the product objects of every other source
# into build_dbg/lib_NAME.so; run a tool against it with  SERENADE_AMD_LIB=$PWD/build_dbg/lib_NAME.so python tools/...
#
# Variants used for DESIGN.md section 6:
#   tools/dbg_variants.sh SAME    conv_gemm.hip -DSRN_DEBUG_SAMETILE                  (every block reads tile (0,0))
#   tools/dbg_variants.sh NOMFMA  conv_gemm.hip -DSRN_DEBUG_SAMETILE -DSRN_DBG_NOMFMA
#   tools/dbg_variants.sh NOSTORE conv_gemm.hip -DSRN_DEBUG_SAMETILE -DSRN_DBG_NOSTORE
#   tools/dbg_variants.sh NOLOAD  conv_gemm.hip -DSRN_DEBUG_SAMETILE -DSRN_DBG_NOLOAD
#   tools/dbg_variants.sh TIMING  conv_fast.hip -DSRN_DBG_TIMING                      (tools/looptime.py)
#   tools/dbg_variants.sh M16     conv_fast.hip -DSRN_DBG_MFMA16                      (16x16x32 MFMA issue, wrong math)
#   tools/dbg_variants.sh APL     conv_fast.hip -DSRN_DBG_APL                         (A operand as planes)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -c "from serenade_amd import build; build.build(verbose=False)"
mkdir -p build_dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Iinclude -Iserenade_amd/csrc "$@" \
  -c serenade_amd/csrc/$src -o build_dbg/${src}_$name.o
objs=""
for o in serenade_amd/build/*.o; do
  case "$o" in *"/$src.o") ;; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_dbg/lib_$name.so build_dbg/${src}_$name.o $objs
ls -la build_dbg/lib_$name.so
