#!/bin/bash
# A/B of one source file on ONE GPU box (boxes differ by 1-3 %: numbers from two gpurun calls are not comparable).
# Here:   cp <file> tools/probes/_ab_old.txt   (the OLD version), edit <file>, rebuild;
#         gpurun -- 'bash tools/ab_on_box.sh normalizing-flows-pytorch_amd/csrc/conv_chain.hip c4'
# The box runs the bench twice with the new build, puts the old file back, rebuilds (hipcc is in the image) and runs it twice again.
F=$1; C=${2:-c4}
for i in 1 2; do timeout 300 python bench.py --config $C --skip-cpu 2>&1 | tail -1 | cut -c90-200; done
cp tools/probes/_ab_old.txt $F
python -c 'import importlib; importlib.import_module("normalizing-flows-pytorch_amd._build").build()'
echo "--- old"
for i in 1 2; do timeout 300 python bench.py --config $C --skip-cpu 2>&1 | tail -1 | cut -c90-200; done
