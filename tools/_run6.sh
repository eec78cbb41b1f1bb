cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
python bench.py --config c1 --skip-cpu --steps 100 > gpurun_out/r3g/bench_c1.json 2> gpurun_out/r3g/bench_c1.err
python bench.py --config c2 --skip-cpu --steps 50 > gpurun_out/r3g/bench_c2.json 2> gpurun_out/r3g/bench_c2.err
python -m pytest tests/test_gpu_fused.py -x -q > gpurun_out/r3g/pytest_fused.log 2>&1
