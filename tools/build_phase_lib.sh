#!/bin/bash
# Diagnostic library with in-kernel phase stamps in conv3x3_dma_kernel -> semantic_segmentation_amd/libgsseg_hip_phase.so
set -e
cd "$(dirname "$0")/../semantic_segmentation_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-function -ffp-contract=fast -DGS_C3_PHASE_TIMING -c conv3x3_dma.hip -o build/conv3x3_dma_phase.o
OBJS=$(ls build/*.o | grep -v conv3x3_dma.o | grep -v conv3x3_dma_phase.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libgsseg_hip_phase.so $OBJS build/conv3x3_dma_phase.o
echo built ../libgsseg_hip_phase.so
