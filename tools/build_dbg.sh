#!/bin/bash
# builds the product library and the phase-clock debug variant (tools/_dbg/libmjx_clock.so)
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" 
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -Wno-unused-value -ldl -mllvm -amdgpu-mfma-vgpr-form -DMJX_PHASE_CLOCK -o tools/_dbg/libmjx_clock.so mjrl_amd/csrc/mjx.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -Wno-unused-value -ldl -mllvm -amdgpu-mfma-vgpr-form -DMJX_PFIT_CLOCK -o tools/_dbg/libmjx_pfit.so mjrl_amd/csrc/mjx.hip
