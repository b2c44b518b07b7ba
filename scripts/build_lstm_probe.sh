#!/bin/bash
# Builds scripts/lstm_probe (stage timing of the persistent LSTM kernels); needs the library objects (make -C .../csrc).
set -e
cd "$(dirname "$0")/.."
C=curriculum-learning-for-vln_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -Iinclude -I$C \
  -c scripts/lstm_probe.hip -o /tmp/lstm_probe.o "$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/lstm_probe.o $C/api.o $C/gemm.o $C/attention.o $C/pointwise.o $C/envdrop.o \
  $C/features.o $C/optim.o $C/monitor.o $C/follower.o $C/bn_mlp.o -o scripts/lstm_probe
