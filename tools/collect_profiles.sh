#!/bin/bash
# Round profile collection on the GPU box: kernel-trace summaries of the bench command per config, PMC passes (separate runs,
# --kernel-trace only) of the dominant kernels, the asymptotic kernel sweep and the per-config bench lines -> gpurun_out/prof_rNN/.
#   bash tools/collect_profiles.sh r02
set -u
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in c4 c1 c2 c3 c5 rnvp_img; do
  rm -rf /tmp/ks_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -o st -- python $GRAFT_REPO_ROOT/bench.py --config $c --skip-cpu --steps 20 > $OUT/${R}_rocprof_bench_$c.json 2> /dev/null
  F=$(find /tmp/ks_$c -name "st_kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -40 $F > $OUT/${R}_${c}_kernel_stats.csv
done
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o fetch -- python $GRAFT_REPO_ROOT/tools/pmc_round.py c4 c1 c2 c3 c5 fpp_img rnvp_img c4:512 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o write -- python $GRAFT_REPO_ROOT/tools/pmc_round.py c4 c1 c2 c3 c5 fpp_img rnvp_img c4:512 > $OUT/pmc_write.log 2>&1
FC=$(find /tmp/pmc_f -name "fetch_counter_collection.csv" | head -1)
WC=$(find /tmp/pmc_w -name "write_counter_collection.csv" | head -1)
cd $GRAFT_REPO_ROOT
NF_PMC_CONFIGS=c4,c1,c2,c3,c5,fpp_img,rnvp_img,c4:512 python tools/pmc_round.py --json $FC $WC $OUT/${R}_pmc.json > $OUT/pmc_json.log 2>&1
cp $OUT/${R}_pmc.json $GRAFT_REPO_ROOT/profiles/${R}_pmc.json   # (the bench lines below read the round's PMC file from profiles/)
# the image Flow++ path (row f4): kernel stats of its bench command, its bench line, per-launch device times, fused vs module stack
cd /tmp
rm -rf /tmp/ks_f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_f -o st -- python $GRAFT_REPO_ROOT/bench.py --config fpp_img --skip-cpu --steps 20 > $OUT/${R}_rocprof_bench_fpp_img.json 2> /dev/null
F=$(find /tmp/ks_f -name "st_kernel_stats.csv" | head -1)
[ -n "$F" ] && head -45 $F > $OUT/${R}_fpp_img_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python bench.py --config fpp_img --steps 50 --cpu-seconds 15 > $OUT/${R}_bench_fpp_img.json 2> /dev/null
python tools/probes/flowpp_img_kernels.py 64 > $OUT/${R}_fpp_img_launch_times.txt 2>&1
python tools/probes/flowpp_img_step.py 2 64 10 > $OUT/${R}_fpp_img_step.txt 2>&1
python tools/kernel_sweep.py > $OUT/${R}_kernel_sweep.txt 2> $OUT/sweep.err
for c in c1 c2 c3 c4 c5 rnvp_img; do
  python bench.py --config $c --steps 50 --cpu-seconds 15 > $OUT/${R}_bench_$c.json 2> /dev/null
done
python bench.py > $OUT/${R}_bench_default.json 2> $OUT/bench_default.err
# per-step launch census of the two metric configs (steady-state window between marker launches) and per-grid chain launch times
cd /tmp
for c in c4 c1 fpp_img rnvp_img; do
  rm -rf /tmp/sk_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/sk_$c -o st -- python $GRAFT_REPO_ROOT/tools/step_kernels.py $c > /dev/null 2> $OUT/sk_$c.err
  T=$(find /tmp/sk_$c -name "st_kernel_trace.csv" | head -1)
  TOP=45 python $GRAFT_REPO_ROOT/tools/step_kernels.py --census $T > $OUT/${R}_${c}_step_kernels.txt
  python $GRAFT_REPO_ROOT/tools/step_kernels.py --by-grid $T chain >> $OUT/${R}_${c}_step_kernels.txt
  python $GRAFT_REPO_ROOT/tools/step_kernels.py --by-grid $T head >> $OUT/${R}_${c}_step_kernels.txt
done
# config 4's literal batch (512) on one GPU: per-step census (large-batch kernels of csrc/conv_bulk.hip), and the sampling pass of the
# image Flow++ object (inverse census: the forward / inverse gap of round 3)
rm -rf /tmp/sk_b512 /tmp/fi_inv
NF_BATCH=512 NF_STEPS=4 rocprofv3 --kernel-trace --output-format csv -d /tmp/sk_b512 -o st -- python $GRAFT_REPO_ROOT/tools/step_kernels.py c4 > /dev/null 2> $OUT/sk_b512.err
T=$(find /tmp/sk_b512 -name "st_kernel_trace.csv" | head -1)
NF_STEPS=4 TOP=45 python $GRAFT_REPO_ROOT/tools/step_kernels.py --census $T > $OUT/${R}_c4_b512_step_kernels.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/fi_inv -o st -- python $GRAFT_REPO_ROOT/tools/probes/fpp_img_inverse.py 5 > $OUT/fpp_inv.log 2>&1
T=$(find /tmp/fi_inv -name "st_kernel_trace.csv" | head -1)
NF_STEPS=5 TOP=30 python $GRAFT_REPO_ROOT/tools/step_kernels.py --census $T > $OUT/${R}_fpp_img_inverse_kernels.txt
grep -E "round trip|pass:" $OUT/fpp_inv.log >> $OUT/${R}_fpp_img_inverse_kernels.txt
cd $GRAFT_REPO_ROOT
# large-batch kernels in isolation (conv_bulk.hip): forward / data gradient per layer, weight gradient per 16 layers, and its phase stamps
python tools/probes/bulk_time.py 512 16 > $OUT/${R}_conv_bulk_times.txt 2>&1
python tools/probes/bulk_time.py 512 8 >> $OUT/${R}_conv_bulk_times.txt 2>&1
for h in 16 8; do python tools/probes/wgrad_time.py $h 512 16 >> $OUT/${R}_conv_bulk_times.txt 2>&1; done
# C1: the one-workgroup kernels (flow_solo.hip) in their settings, same box
for m in 0 1 2 3; do echo "NF_FLOW_SOLO=$m" >> $OUT/${R}_c1_solo_modes.txt; NF_FLOW_SOLO=$m python bench.py --config c1 --skip-cpu --steps 50 | cut -c1-330 >> $OUT/${R}_c1_solo_modes.txt; done
# model-level asymptotic sweep: CIFAR-shape Glow at B = 512, 2048 per GPU; 2-D models up to 2^22 rows
python tools/model_sweep.py > $OUT/${R}_model_sweep.txt 2> $OUT/model_sweep.err
# the reference path's own spread under row permutations and the per-step error profile of C1; the kink census runs on the host
python tools/probes/parity_depth.py c1 > $OUT/${R}_c1_parity_depth.txt 2>&1
python tools/cpu_threads.py c4 8 16 32 64 > $OUT/${R}_cpu_threads.txt 2>&1
SECONDS_PER=4 python tools/cpu_threads.py c1 1 4 8 16 32 >> $OUT/${R}_cpu_threads.txt 2>&1
# the data-parallel control flow on REAL RCCL with a one-rank group (NF_DP_FORCE_COLLECTIVE=1: barriers, start-up broadcast, graph A + eager
# all-reduce + graph B, or -- NF_DP_ONE_GRAPH=1 -- the all-reduce captured inside the step graph) against the plain single-process step
for c in c4 c1 c5; do for m in "NF_DP_FORCE_COLLECTIVE=0" "NF_DP_FORCE_COLLECTIVE=1" "NF_DP_FORCE_COLLECTIVE=1 NF_DP_ONE_GRAPH=1"; do
  echo "== $c $m" >> $OUT/${R}_dp_one_rank.txt
  env $m python bench.py --config $c --skip-cpu --steps 50 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('samples/s', d['value'], 'ms/step', d['ms_per_step'], 'event median', d.get('ms_per_step_event_median'), 'one_graph', d['config'].get('dp_one_graph'), d['config'].get('collective'))" >> $OUT/${R}_dp_one_rank.txt 2>&1
done; done
# issue rate of the transcendental VALU instructions (the roof quoted for the mixture-of-logistics kernels)
hipcc --offload-arch=gfx950 -O3 -o /tmp/vexp_rate_probe tools/probes/vexp_rate_probe.hip 2> /dev/null && /tmp/vexp_rate_probe > $OUT/${R}_vexp_rate.txt 2>&1
# run-to-run reproducibility of the train step: the racing fast path and the ordered mode (csrc/nf_det.h)
rm -f gpurun_out/determinism.txt
NF_DETERMINISTIC=0 python tools/determinism_probe.py c1 c2 c3 c4 c5 rnvp_img fpp_img c4_b512 > $OUT/det0.log 2>&1
NF_DETERMINISTIC=1 python tools/determinism_probe.py c1 c2 c3 c4 c5 rnvp_img fpp_img c4_b512 > $OUT/det1.log 2>&1
cp gpurun_out/determinism.txt $OUT/${R}_determinism.txt
# what the ordered mode costs on the two metric configs
for c in c4 c1; do for m in 0 1; do echo "== $c NF_DETERMINISTIC=$m" >> $OUT/${R}_deterministic_cost.txt; NF_DETERMINISTIC=$m python bench.py --config $c --skip-cpu --steps 50 2> /dev/null | cut -c1-200 >> $OUT/${R}_deterministic_cost.txt; done; done
python tools/probes/solo_time.py > $OUT/${R}_c1_solo_kernel_times.txt 2>&1
python tools/probes/host_step_cost.py > $OUT/${R}_c1_host_step_cost.txt 2>&1
python bench.py --config c4 --scaling strong --skip-cpu --steps 20 > $OUT/${R}_bench_c4_strong_one_rank.json 2> /dev/null
python bench.py --config fpp_img --steps 20 --skip-cpu > $OUT/${R}_bench_fpp_img_skipcpu.json 2> /dev/null
# image Flow++: weight gradients per coupling / deferred with several slab rules; the exchange probe (why XCD placement does not help)
python tools/probes/fpp_img_wgrad_rounds.py 32 64 10 2>&1 | grep -v amdgpu > $OUT/${R}_fpp_img_wgrad_defer.txt
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/xcd_probe tools/probes/xcd_exchange_probe.hip 2> /dev/null && timeout 300 /tmp/xcd_probe > $OUT/${R}_xcd_exchange_probe.txt 2>&1
# head backward in two parts: whole / deferred, on the two configs that have heads
python - > $OUT/${R}_head_params_defer.txt 2>&1 <<'PYEOF'
import importlib, subprocess, sys, json
for cfg, extra in (('c4', []), ('c4', ['--batch', '512']), ('fpp_img', [])):
    for on in (1, 0):
        code = ("import importlib,sys; sys.argv=['bench.py','--config','%s','--skip-cpu','--steps','20'%s]; "
                "F=importlib.import_module('normalizing-flows-pytorch_amd.functional'); F.HEAD_PARAMS_DEFER=bool(%d); "
                "import runpy; runpy.run_path('bench.py', run_name='__main__')") % (cfg, ''.join(",'%s'" % e for e in extra), on)
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print('%-8s %-14s HEAD_PARAMS_DEFER=%d  %9.1f samples/s  %8.3f ms/step' % (cfg, ' '.join(extra), on, d['value'], d['ms_per_step']))
        except Exception as e:
            print(cfg, extra, on, 'failed', r.stderr[-300:])
PYEOF
python -m pytest tests/test_gpu_fullsize_parity.py -q > $OUT/pytest_fullsize.log 2>&1
cp gpurun_out/fullsize_parity.txt $OUT/${R}_fullsize_parity.txt
ls -la $OUT
