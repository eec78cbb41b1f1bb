cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
python tools/probes/chain_prof.py --build > gpurun_out/r3h/build.log 2>&1
python tools/probes/chain_prof.py 96 192 4 4 64 --cpl --bwd > gpurun_out/r3h/prof_4x4_bwd.txt 2>&1
python tools/probes/chain_prof.py 24 48 8 8 64 --cpl --bwd > gpurun_out/r3h/prof_8x8_bwd.txt 2>&1
python tools/probes/chain_prof.py 6 12 16 16 64 --cpl --bwd > gpurun_out/r3h/prof_16x16_bwd.txt 2>&1
