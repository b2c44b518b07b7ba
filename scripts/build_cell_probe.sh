#!/bin/bash
# Builds scripts/cell_probe (fused LSTM cell vs gemm_nt + lstm_pw); needs the library objects (make -C .../csrc).
set -e
cd "$(dirname "$0")/.."
C=curriculum-learning-for-vln_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -Iinclude -I$C \
  -c scripts/cell_probe.hip -o /tmp/cell_probe.o "$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/cell_probe.o $C/api.o $C/gemm.o $C/attention.o $C/pointwise.o $C/envdrop.o $C/encoder.o \
  $C/features.o $C/optim.o $C/monitor.o $C/follower.o $C/bn_mlp.o -o scripts/cell_probe
