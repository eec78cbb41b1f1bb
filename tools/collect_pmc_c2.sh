set -u
R=r04
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o fetch -- python $GRAFT_REPO_ROOT/tools/pmc_round.py c4 c1 c2 c3 c5 fpp_img c4:512 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o write -- python $GRAFT_REPO_ROOT/tools/pmc_round.py c4 c1 c2 c3 c5 fpp_img c4:512 > $OUT/pmc_write.log 2>&1
FC=$(find /tmp/pmc_f -name "fetch_counter_collection.csv" | head -1)
WC=$(find /tmp/pmc_w -name "write_counter_collection.csv" | head -1)
cd $GRAFT_REPO_ROOT
NF_PMC_CONFIGS=c4,c1,c2,c3,c5,fpp_img,c4:512 python tools/pmc_round.py --json $FC $WC $OUT/${R}_pmc.json > $OUT/pmc_json.log 2>&1
cp $OUT/${R}_pmc.json $GRAFT_REPO_ROOT/profiles/${R}_pmc.json
cp $OUT/${R}_pmc.json $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_new.json
cd /tmp; rm -rf /tmp/ks_c2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_c2 -o st -- python $GRAFT_REPO_ROOT/bench.py --config c2 --skip-cpu --steps 20 > $OUT/${R}_rocprof_bench_c2.json 2> /dev/null
F=$(find /tmp/ks_c2 -name "st_kernel_stats.csv" | head -1)
[ -n "$F" ] && head -40 $F > $OUT/${R}_c2_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python bench.py --config c2 --steps 200 --cpu-seconds 15 > $OUT/${R}_bench_c2.json 2> /dev/null
python bench.py > $OUT/${R}_bench_default.json 2> $OUT/bench_default.err
