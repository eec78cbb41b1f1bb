#!/bin/bash
# Weight-gradient launches of the image conditioners on a side stream next to the persistent chain (NF_CONV_OVERLAP=1), with the launches
# capped at fewer workgroups than compute units (NF_CONV_WGRAD_BLOCKS) so that the chain's whole-compute-unit workgroups never wait for
# them: one box, bench.py --config c4, samples/s | ms per step | event median.
#   gpurun -- 'bash tools/probes/overlap_ab.sh > gpurun_out/overlap_ab.txt'
C=${1:-c4}
run() {
  echo -n "$* : "
  env "$@" python bench.py --config $C --skip-cpu --steps 30 2> /tmp/ab_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_event_median'), 'graph', d['config'].get('hipgraph'))" || tail -3 /tmp/ab_err.txt
}
run NF_CONV_OVERLAP=0
run NF_CONV_OVERLAP=1
run NF_CONV_OVERLAP=1 NF_CONV_WGRAD_BLOCKS=128
run NF_CONV_OVERLAP=1 NF_CONV_WGRAD_BLOCKS=96
run NF_CONV_OVERLAP=1 NF_CONV_WGRAD_BLOCKS=64
run NF_CONV_OVERLAP=1 NF_CONV_WGRAD_BLOCKS=128 NF_CONV_OFFLOAD_MIN=8
run NF_CONV_OVERLAP=0
