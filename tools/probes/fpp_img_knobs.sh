#!/bin/bash
# knobs of the image Flow++ conditioner on one box (bench.py --config fpp_img [--batch B]): samples/s | ms per step | event median
B=${1:-64}
run() { echo -n "B=$B $* : "; env "$@" python bench.py --config fpp_img --batch $B --skip-cpu --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_event_median'))"; }
run X=0
for t in 257 513 1000 2049 100000; do run NF_FLOWPP_IMG_TILE64=$t; done
run X=0
