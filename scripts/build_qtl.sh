#!/bin/bash
# ticket timeline build of k_step4_queue (never shipped): scripts/timeline_queue.py
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -disable-machine-licm -DRMJ_QTL \
  -Wno-unused-result -Wno-unused-value riichienv_amd/csrc/rmj_api.hip -o riichienv_amd/libriichi_mi355x_qtl.so
