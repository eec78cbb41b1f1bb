run() { echo -n "$* : "; env "$@" python bench.py --config c4 --skip-cpu --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_step_event_median'))"; }
run X=0
run NF_CONV_WGRAD_BLOCKS=512
run NF_CONV_WGRAD_BLOCKS=384
run NF_CONV_WGRAD_BLOCKS=192
run NF_GLOW_HEAD_BWD_TP=128
run X=0
