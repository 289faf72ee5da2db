#!/bin/bash
# Dev-only: build conv_igemm variants into separate libs and time them.
set -e
cd /root/repo/tensorflow_ocr_amd/csrc
for wps in 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DOCR_WPS=$wps -c conv_igemm.hip -o build/conv_igemm_w$wps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libocr_hip_w$wps.so build/conv_igemm_w$wps.o build/abi.o
done
