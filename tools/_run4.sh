cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
python -m pytest tests/test_gpu_convnet.py -x -q > gpurun_out/r3i/pytest_convnet.log 2>&1
python -m pytest tests/test_gpu_models.py -x -q -k "img or image" > gpurun_out/r3i/pytest_models_img.log 2>&1
python bench.py --config c4 --skip-cpu > gpurun_out/r3i/bench_c4.json 2> gpurun_out/r3i/bench_c4.err
NF_CONV_TILE64=0 python bench.py --config c4 --skip-cpu > gpurun_out/r3i/bench_c4_t128.json 2> gpurun_out/r3i/bench_c4_t128.err
python tools/probes/chain_prof.py --build > gpurun_out/r3i/build.log 2>&1
python tools/probes/chain_prof.py 96 192 4 4 64 --cpl --bwd > gpurun_out/r3i/prof_4x4.txt 2>&1
python tools/probes/chain_prof.py 24 48 8 8 64 --cpl --bwd > gpurun_out/r3i/prof_8x8.txt 2>&1
