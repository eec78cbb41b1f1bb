#!/bin/bash
# Phase stamps of workgroup 0 of the persistent ConvNet chain kernel at the three levels of config 4 (B = 64): forward and backward,
# with the coupling in the launch (the launch a model makes).  VERDICT r05 item 4(b): "a committed table that accounts for every us of a layer".
#   bash tools/chain_phase_table.sh > gpurun_out/r06_chain_phases.txt
cd $GRAFT_REPO_ROOT
python tools/probes/chain_prof.py --build > /dev/null 2>&1
for shape in "6 12 16 16" "24 48 8 8" "96 192 4 4"; do
  echo "=== level: $shape  (I O H W), B = 64, coupling in the launch, forward + backward"
  python tools/probes/chain_prof.py $shape 64 --cpl --bwd $HEAD 2>&1 | grep -v Warning | grep -v amdgpu.ids
done
