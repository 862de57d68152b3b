#!/bin/bash
# tools/build_variant.sh <name> <source.hip> [-D...]: liblsi_hip_<name>.so = the
# current objects with <source.hip> rebuilt under extra flags (select it with
# LSI_HIP_LIB=<name>).  Experiment builds for A/B runs on the GPU box.
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../layered-scene-inference_amd"
obj=csrc/${src%.hip}.$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
  -munsafe-fp-atomics -fno-fast-math -Wno-unused-function "$@" -c csrc/$src -o $obj
objs=""
for s in lsi_splat lsi_splat_stream lsi_splat_stream2 lsi_splat_tile lsi_splat_bwd_stream lsi_splat_sweep lsi_sampling lsi_loss lsi_bn lsi_host lsi_conv lsi_conv_wgrad lsi_conv_igemm lsi_conv_wgrad_igemm lsi_conv_first; do
  if [ "$s.hip" == "$src" ]; then objs="$objs $obj"; else objs="$objs csrc/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblsi_hip_$name.so $objs
echo liblsi_hip_$name.so
