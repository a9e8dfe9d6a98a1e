#!/bin/bash
# usage: variant.sh name "-DIPP_NT_STORES=0 ..."  -> builds tools/probes/libipp_$name.so and prints the resource usage of k_step_patch<2>
cd "$(dirname "$0")/../ipp-rl_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm $2 -shared ipp_engine.hip -o ../../tools/probes/libipp_$1.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "k_step_patchILi2" | grep -E "VGPRs:|Spill|ScratchSize|Occupancy" | tr '\n' ' '
echo " <- $1"
