#!/bin/bash
# instruction accounting build of the library (PROF marks end the wave on request); never loaded by the product path
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -disable-machine-licm -DRMJ_CUTS \
  -Wno-unused-result -Wno-unused-value riichienv_amd/csrc/rmj_api.hip -o riichienv_amd/libriichi_mi355x_cuts.so
