set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --cpu-seconds 12 > $OUT/bench_default.json 2> $OUT/bench_default.err
python -m pytest tests/test_gpu_fullsize_parity.py -x -q -k "c1 or c2 or c5 or kink or cifar_shape" > $OUT/pytest_parity.log 2>&1
cp gpurun_out/fullsize_parity.txt $OUT/ 2>/dev/null
python -m pytest tests/test_gpu_models.py tests/test_gpu_datagen.py -x -q > $OUT/pytest_models.log 2>&1
python tools/probes/copy_hunt.py c1 > $OUT/copy_hunt_c1.txt 2>&1
python tools/probes/copy_hunt.py c4 > $OUT/copy_hunt_c4.txt 2>&1
