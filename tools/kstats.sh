#!/bin/bash
# Register / scratch / occupancy of every traversal kernel for a set of build knobs (compile only, no GPU): tools/kstats.sh [-DKNOB=v ...]
# also leaves the assembly in /tmp/kstats.s
cd "$(dirname "$0")/../path_tracer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wno-unused-parameter -Wno-missing-field-initializers "$@" -S --cuda-device-only -o /tmp/kstats.s -x hip pt_kernels.hip -Rpass-analysis=kernel-resource-usage 2>/tmp/kstats.log
grep -E "Function Name|  VGPRs:|SGPRs Spill|ScratchSize|Occupancy" /tmp/kstats.log | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - - \
  | sed 's/Function Name: _ZN2pt12_GLOBAL__N_1//; s/EvNS[^\t]*//; s/EvNS_9SceneView[^\t]*//' | grep -E "${KSTATS_FILTER:-closest|any|fused}"
