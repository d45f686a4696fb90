#!/bin/bash
# timeline build of k_step4 (never shipped): scripts/timeline4.py
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DRMJ_TL4 \
  -Wno-unused-result -Wno-unused-value riichienv_amd/csrc/rmj_api.hip -o riichienv_amd/libriichi_mi355x_tl4.so
