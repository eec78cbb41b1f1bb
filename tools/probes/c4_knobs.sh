#!/bin/bash
# switches of the image-conditioner path on one box (bench.py --config c4): samples/s | ms per step | event median
run() { echo -n "$* : "; env "$@" python bench.py --config c4 --skip-cpu --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_event_median'))"; }
run X=0
run NF_CONV_BULK_MIN_PX=4097
run NF_CONV_BULK_MIN_PX=1025
run NF_CONV_WGRAD_FROM_STORE=0
run NF_CONV_HALO=0
run X=0
