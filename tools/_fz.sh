cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03fz
PYTHONPATH=.:tests timeout 3000 python tools/fuzz_parity.py 3000 300000 > gpurun_out/r03fz/parity.txt 2>&1; tail -2 gpurun_out/r03fz/parity.txt
PYTHONPATH=.:tests timeout 1200 python tools/fuzz_fe.py 250 60000 > gpurun_out/r03fz/fe.txt 2>&1; tail -1 gpurun_out/r03fz/fe.txt
GDMIX_FE_HOT_MIN=300 GDMIX_FE_WINDOW_BITS=11 GDMIX_FE_CHUNK=3001 PYTHONPATH=.:tests timeout 900 python tools/fuzz_fe.py 80 70000 > gpurun_out/r03fz/fe_hooks.txt 2>&1; tail -1 gpurun_out/r03fz/fe_hooks.txt
