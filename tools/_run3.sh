set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_convnet.py -x -q > $OUT/pytest_convnet.log 2>&1
python -m pytest tests/test_gpu_models.py -x -q -k "img or image" > $OUT/pytest_models_img.log 2>&1
python bench.py --config c4 --skip-cpu > $OUT/bench_c4.json 2> $OUT/bench_c4.err
