#!/bin/bash
# Builds scripts/xcd_probe (one XCD's streaming bandwidth and barrier latency; stand-alone)
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/xcd_probe.hip -o scripts/xcd_probe "$@"
