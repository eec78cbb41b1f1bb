#!/bin/bash
# whole-flow launch (NF_GLOW_FLOW=1) against per-step launches with the deferred fold (NF_GLOW_FLOW=steps) by batch: where 'auto' should switch
run() { c=$1; b=$2; shift 2; echo -n "$c B=$b $* : "; env "$@" python bench.py --config $c --batch $b --skip-cpu --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_event_median'))"; }
for b in 2048 4096 8192 16384; do run c2 $b NF_GLOW_FLOW=1; run c2 $b NF_GLOW_FLOW=steps; done
for b in 2048 4096 16384; do run c1 $b NF_GLOW_FLOW=1; run c1 $b NF_GLOW_FLOW=steps; done
