import os, sys, time, tempfile, shutil
sys.path.insert(0, ".")
import numpy as np
from gdmix_amd import gdmix as cli, model as M, synthetic, solver as S
from gdmix_amd.partition_dirs import write_partition_dir
import logging; logging.disable(logging.INFO)
b = synthetic.make_survey_batch(200000, 32, 8, 65536, seed=synthetic.C5_SEED, size_dist="c5zipf", with_uid=True)
T0 = [0.0]
def wrap(owner, name, label):
    fn = getattr(owner, name)
    def w(*a, **k):
        t = time.perf_counter()
        try: return fn(*a, **k)
        finally: print(f"   {(t-T0[0])*1e3:8.1f} -> {(time.perf_counter()-T0[0])*1e3:8.1f} ms  {label}", flush=True)
    setattr(owner, name, w)
wrap(S.REDeviceSolver, "score", "solver.score")
wrap(S.REDeviceSolver, "pack", "solver.pack")
wrap(S.REDeviceSolver, "solve", "solver.solve")
wrap(S.REDeviceSolver, "__init__", "REDeviceSolver()")
wrap(M, "host_array", "host_array")
wrap(M.RandomEffectLRLBFGSModel, "_write_behind", "_write_behind")
wrap(M.RandomEffectLRLBFGSModel, "_predict", "_predict")
wrap(M.RandomEffectLRLBFGSModel, "_solve_batch", "_solve_batch")
wrap(M.RandomEffectLRLBFGSModel, "_read", "_read")
wrap(S.SolveResult, "to_host", "to_host")
with tempfile.TemporaryDirectory() as d:
    argv, members, _ = write_partition_dir(d, b, 8, 65536)
    for rep in range(3):
        shutil.rmtree(os.path.join(d, "models"), ignore_errors=True); shutil.rmtree(os.path.join(d, "ts"), ignore_errors=True)
        print("=== run", rep, flush=True)
        T0[0] = time.perf_counter()
        cli.run(argv)
        print(f"   total {(time.perf_counter()-T0[0])*1e3:.1f} ms", flush=True)
