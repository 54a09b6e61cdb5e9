#!/bin/bash
# developer build of a variant of the kernel library: tools/build_variant.sh NAME "-DVD_PD=4 ..." -> build_dbg/libviddet_NAME.so
# (only vd_conv.hip is recompiled with the extra flags; load it with VD_LIB=build_dbg/libviddet_NAME.so).  Not used by the product.
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../viddet_amd/csrc"
mkdir -p ../../build_dbg
F="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -fno-slp-vectorize -I../../include $FLAGS"
/opt/rocm/bin/hipcc $F -c vd_conv.hip -o ../../build_dbg/vd_conv_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_dbg/libviddet_$NAME.so ../../build_dbg/vd_conv_$NAME.o \
    vd_conv_sk.o vd_conv_par.o vd_wgrad_halo.o vd_conv_bf16.o vd_conv_bf16_sk.o vd_conv_c32_bf16.o vd_stem.o vd_bn.o vd_pointwise.o vd_yolo.o vd_api.o
