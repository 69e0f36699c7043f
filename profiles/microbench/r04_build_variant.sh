#!/bin/bash
# usage: r04_build_variant.sh NAME "-DFOO -DBAR"   -> build/ab/libscaldpc_NAME.so: the library with the bp translation unit rebuilt under extra
# defines, for A/B runs of compile-time variants (SCALDPC_SO=build/ab/libscaldpc_NAME.so python bench.py ...; build/ is not in history)
set -e
name=$1; defs=$2
out=/root/repo/build/ab; mkdir -p $out /tmp/var_$name
cd /root/repo/sca-ldpc_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-pass-failed $defs -c -o /tmp/var_$name/scaldpc_bp.o scaldpc_bp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $out/libscaldpc_$name.so scaldpc_runtime.o /tmp/var_$name/scaldpc_bp.o scaldpc_qary.o
ls -la $out/libscaldpc_$name.so
