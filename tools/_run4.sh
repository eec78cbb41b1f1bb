cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
python -m pytest tests/test_gpu_convnet.py -x -q > gpurun_out/r3j/pytest_convnet.log 2>&1
python -m pytest tests/test_gpu_models.py -x -q -k "img or image" > gpurun_out/r3j/pytest_models_img.log 2>&1
python bench.py --config c4 --skip-cpu > gpurun_out/r3j/bench_c4.json 2> gpurun_out/r3j/bench_c4.err
