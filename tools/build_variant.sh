#!/bin/bash
# build_variant.sh NAME SOURCE.hip -DFLAG...: compile one source with extra flags, link it with the regular objects of the other
# sources (tntorch_amd/csrc/build/*.o from `python __graft_entry__.py`) into tntorch_amd/libttround_NAME.so; use it with
# TTR_LIB_PATH=tntorch_amd/libttround_NAME.so.
set -e
cd "$(dirname "$0")/../tntorch_amd/csrc"
name=$1; src=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$src" -o "build/${src%.hip}_$name.o"
objs=""
for f in ttr_api ttr_gemm ttr_qr ttr_eigh ttr_cp ttr_sweep ttr_bjacobi ttr_eigsel ttr_roundtt; do
  if [ "$f.hip" == "$src" ]; then objs="$objs build/${f}_$name.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "../libttround_$name.so"
echo "built tntorch_amd/libttround_$name.so"
