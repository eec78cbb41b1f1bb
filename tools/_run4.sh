cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
python -m pytest tests/test_gpu_convnet.py -x -q > gpurun_out/r3l/pytest_convnet.log 2>&1
python bench.py --config c4 --skip-cpu > gpurun_out/r3l/bench_c4.json 2> gpurun_out/r3l/bench_c4.err
