# forward launch with and without the step's head in its prologue: phase stamps of workgroup 0 (tools/probes/chain_prof.py)
cd $GRAFT_REPO_ROOT
python tools/probes/chain_prof.py --build > /dev/null 2>&1
for shape in "24 48 8 8" "96 192 4 4"; do
  for h in "" "--head"; do
    echo "=== level: $shape  (I O H W), B = 64, coupling in the launch, forward $h"
    python tools/probes/chain_prof.py $shape 64 --cpl $h 2>&1 | grep -E "total|prologue|layer 0"
  done
done
