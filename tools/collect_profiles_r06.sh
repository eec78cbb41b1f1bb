#!/bin/bash
# Round-6 profile collection on the GPU box (a trimmed tools/collect_profiles.sh: the probes whose subject did not change this round are not re-run).
#   bash tools/collect_profiles_r06.sh     -> gpurun_out/prof_r06/
set -u
R=r06
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in c4 c1 c2 c3 c5 rnvp_img fpp_img; do
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
python tools/cpu_threads.py c4 8 16 32 64 > $OUT/${R}_cpu_threads.txt 2>&1
SECONDS_PER=4 python tools/cpu_threads.py c1 1 4 8 16 32 >> $OUT/${R}_cpu_threads.txt 2>&1
cp $OUT/${R}_cpu_threads.txt $GRAFT_REPO_ROOT/profiles/${R}_cpu_threads.txt
python tools/kernel_sweep.py > $OUT/${R}_kernel_sweep.txt 2> $OUT/sweep.err
python tools/model_sweep.py > $OUT/${R}_model_sweep.txt 2> $OUT/model_sweep.err
for c in c1 c2 c3 c4 c5 rnvp_img fpp_img; do
  python bench.py --config $c --steps 50 --cpu-seconds 15 > $OUT/${R}_bench_$c.json 2> /dev/null
done
python bench.py > $OUT/${R}_bench_default.json 2> $OUT/bench_default.err
cp gpurun_out/bench_detail.json $OUT/${R}_bench_detail.json
# per-step launch census (steady-state window between marker launches) and per-grid chain launch times
cd /tmp
for c in c4 c1 c3 fpp_img rnvp_img; do
  rm -rf /tmp/sk_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/sk_$c -o st -- python $GRAFT_REPO_ROOT/tools/step_kernels.py $c > /dev/null 2> $OUT/sk_$c.err
  T=$(find /tmp/sk_$c -name "st_kernel_trace.csv" | head -1)
  TOP=45 python $GRAFT_REPO_ROOT/tools/step_kernels.py --census $T > $OUT/${R}_${c}_step_kernels.txt
  python $GRAFT_REPO_ROOT/tools/step_kernels.py --by-grid $T chain >> $OUT/${R}_${c}_step_kernels.txt
done
rm -rf /tmp/sk_b512
NF_BATCH=512 NF_STEPS=4 rocprofv3 --kernel-trace --output-format csv -d /tmp/sk_b512 -o st -- python $GRAFT_REPO_ROOT/tools/step_kernels.py c4 > /dev/null 2> $OUT/sk_b512.err
T=$(find /tmp/sk_b512 -name "st_kernel_trace.csv" | head -1)
NF_STEPS=4 TOP=45 python $GRAFT_REPO_ROOT/tools/step_kernels.py --census $T > $OUT/${R}_c4_b512_step_kernels.txt
cd $GRAFT_REPO_ROOT
# run-to-run reproducibility and the cost of the ordered mode
rm -f gpurun_out/determinism.txt
NF_DETERMINISTIC=0 python tools/determinism_probe.py c1 c2 c3 c4 c5 rnvp_img fpp_img c4_b512 > $OUT/det0.log 2>&1
NF_DETERMINISTIC=1 python tools/determinism_probe.py c1 c2 c3 c4 c5 rnvp_img fpp_img c4_b512 > $OUT/det1.log 2>&1
cp gpurun_out/determinism.txt $OUT/${R}_determinism.txt
for c in c4 c1 c2 c3 c5 rnvp_img fpp_img; do for m in 0 1; do echo "== $c NF_DETERMINISTIC=$m" >> $OUT/${R}_deterministic_cost_all.txt; NF_DETERMINISTIC=$m python bench.py --config $c --skip-cpu --steps 20 2> /dev/null | cut -c1-200 >> $OUT/${R}_deterministic_cost_all.txt; done; done
for m in 0 1; do echo "== c4 batch 512 NF_DETERMINISTIC=$m" >> $OUT/${R}_deterministic_cost_all.txt; NF_DETERMINISTIC=$m python bench.py --config c4 --batch 512 --skip-cpu --steps 5 2> /dev/null | cut -c1-200 >> $OUT/${R}_deterministic_cost_all.txt; done
# the data-parallel control flow on REAL RCCL with a one-rank group
for c in c4 c1; do for m in "NF_DP_FORCE_COLLECTIVE=0" "NF_DP_FORCE_COLLECTIVE=1" "NF_DP_FORCE_COLLECTIVE=1 NF_DP_ONE_GRAPH=1"; do
  echo "== $c $m" >> $OUT/${R}_dp_one_rank.txt
  env $m python bench.py --config $c --skip-cpu --steps 50 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('samples/s', d['value'], 'ms/step', d['ms_per_step'], 'one_graph', d['config'].get('dp_one_graph'), d['config'].get('collective'))" >> $OUT/${R}_dp_one_rank.txt 2>&1
done; done
python bench.py --config c4 --scaling strong --skip-cpu --steps 20 > $OUT/${R}_bench_c4_strong_one_rank.json 2> /dev/null
ls -la $OUT
