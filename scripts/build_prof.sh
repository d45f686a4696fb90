#!/bin/bash
# profiling build of the library (section timers compiled in); never loaded by the product path
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DRMJ_PROFILE \
  -Wno-unused-result -Wno-unused-value riichienv_amd/csrc/rmj_api.hip -o riichienv_amd/libriichi_mi355x_prof.so
