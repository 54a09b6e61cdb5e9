#!/bin/bash
# developer build of the kernel library with in-kernel s_memtime stamps (tools/stamp_conv.py); not used by the product
set -e
cd "$(dirname "$0")/../viddet_amd/csrc"
make
mkdir -p ../../build_dbg
F="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -fno-slp-vectorize -I../../include -DVD_STAMP=1"
/opt/rocm/bin/hipcc $F -c vd_conv.hip -o ../../build_dbg/vd_conv.o &
/opt/rocm/bin/hipcc $F -c vd_conv_bf16.hip -o ../../build_dbg/vd_conv_bf16.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_dbg/libviddet_stamp.so ../../build_dbg/vd_conv.o \
    ../../build_dbg/vd_conv_bf16.o vd_conv_sk.o vd_conv_par.o vd_wgrad_halo.o vd_conv_bf16_sk.o vd_conv_c32_bf16.o vd_stem.o vd_bn.o vd_pointwise.o vd_yolo.o vd_api.o
