#!/bin/bash
# developer tool: where k_nms spends its time.  Builds vd_yolo.hip with -DVD_NMS_PROBE=1..5 (the kernel ends after the select /
# sort / box decode / IoU bitmask / sweep) and times `bench.py --mode detect --dtype bf16` 's NMS record with each library
# (VD_LIB): the differences are the phases.  usage (GPU box): bash tools/nms_probe.sh [batch=32]
set -e
B=${1:-32}
cd "$(dirname "$0")/.."
mkdir -p build_dbg gpurun_out/nms_probe
F="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -fno-slp-vectorize -Iinclude"
for n in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc $F -DVD_NMS_PROBE=$n -c viddet_amd/csrc/vd_yolo.hip -o build_dbg/vd_yolo_p$n.o
  (cd viddet_amd/csrc && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_dbg/libviddet_nmsp$n.so vd_conv.o vd_conv_sk.o vd_conv_par.o \
      vd_wgrad_halo.o vd_conv_bf16.o vd_conv_bf16_sk.o vd_conv_c32_bf16.o vd_stem.o vd_bn.o vd_pointwise.o ../../build_dbg/vd_yolo_p$n.o vd_api.o)
done
for n in 0 1 2 3 4 5; do
  if [ $n = 0 ]; then L=""; else L="build_dbg/libviddet_nmsp$n.so"; fi
  VD_LIB=$L python bench.py --mode detect --dtype bf16 --batch $B --no-cpu-baseline > gpurun_out/nms_probe/b${B}_p$n.json 2> gpurun_out/nms_probe/b${B}_p$n.err
  python - "$n" gpurun_out/nms_probe/b${B}_p$n.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("probe %s: nms %.4f ms  decode %.4f ms" % (sys.argv[1], d["kernels"]["nms"]["ms"], d["kernels"]["decode_filter"]["ms"]), flush=True)
PY
done
