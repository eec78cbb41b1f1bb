#!/bin/bash
# switches of the 2-D launch paths on one box (bench.py --config c1 / c2 / c3 / c5): samples/s | ms per step | event median
run() { c=$1; shift; echo -n "$c $* : "; env "$@" python bench.py --config $c --skip-cpu --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_event_median'))"; }
run c2 X=0
run c2 NF_GLOW_FLOW=1
run c2 NF_GLOW_FLOW_STEPS=0
run c2 NF_GLOW_FLOW=0
run c1 X=0
run c1 NF_FLOW_DEFER_FOLD=0
run c1 NF_GLOW_FLOW=0
run c5 X=0
run c5 NF_MAF_FLOW=0
run c3 X=0
run c3 NF_FLOWPP_DEFER=0
