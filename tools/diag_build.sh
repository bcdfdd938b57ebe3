#!/bin/bash
# Diagnostic variants of the forward MLP kernels (never loaded by the product or the tests): field_mlp.hip rebuilt
# with -DMI_DIAG_SIN=<mode> and linked with the product's other objects into gpurun_tools/libmirender_diag<mode>.so.
# Usage: bash tools/diag_build.sh 1 2 3     then on the GPU box: MI_LIB=gpurun_tools/libmirender_diag1.so python tools/perf_quick.py
set -e
cd "$(dirname "$0")/.."
C=msra-practice-project_amd/csrc
mkdir -p gpurun_tools
for mode in "$@"; do
  case "$mode" in sin*)   # sin<k>: the activation's range-reduction variant k (mi_math.h MI_SIN_VARIANT) -> libmirender_sin<k>.so
    k=${mode#sin}
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMI_SIN_VARIANT=$k -c $C/field_mlp.hip -o /tmp/field_mlp_sin$k.o &&
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/field_mlp_sin$k.o $C/_obj/field_mlp_bwd.o $C/_obj/render_stages.o $C/_obj/eval_stages.o $C/_obj/adam_step.o $C/_obj/api.o -o gpurun_tools/libmirender_sin$k.so &&
      echo "built sin variant $k" ) &
    continue ;;
  esac
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMI_DIAG_SIN=$mode -c $C/field_mlp.hip -o /tmp/field_mlp_diag$mode.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/field_mlp_diag$mode.o $C/_obj/field_mlp_bwd.o $C/_obj/render_stages.o $C/_obj/eval_stages.o $C/_obj/adam_step.o $C/_obj/api.o -o gpurun_tools/libmirender_diag$mode.so &&
    echo "built diag $mode" ) &
done
wait
