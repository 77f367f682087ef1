"""Every call of a CLI run's main path stamped with its start and end (ms since the run began): where the main thread waits.
    python tools/cli_first_partition.py [c5|c2] [entities] [partitions] [warm]
Round 5 found the first partition's scoring pass 50 x slower than the others this way (the writer thread encoding 65 536 feature
names under the interpreter lock)."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, ".")
import numpy as np
from gdmix_amd import gdmix as cli, model as M, synthetic, solver as S
from gdmix_amd.partition_dirs import write_partition_dir
import logging; logging.disable(logging.INFO)
shape = sys.argv[1] if len(sys.argv) > 1 else "c5"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 8
warm = len(sys.argv) > 4 and sys.argv[4] == "warm"
dim = 65536 if shape == "c5" else 1024
b = (synthetic.make_survey_batch(E, 32, 8, dim, seed=synthetic.C5_SEED, size_dist="c5zipf", with_uid=True) if shape == "c5"
     else synthetic.make_batch(E, 16, 4, dim, seed=1))
if b.uid is None:
    b.uid = np.arange(b.N, dtype=np.int64)
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
wrap(M.RandomEffectLRLBFGSModel, "_load_weights", "_load_weights")
wrap(M.RandomEffectLRLBFGSModel, "_start_point", "_start_point")
wrap(M.RandomEffectLRLBFGSModel, "end_pipeline", "end_pipeline")
wrap(M.ModelTable, "update", "ModelTable.update")
wrap(M.ModelTable, "add_chunk", "ModelTable.add_chunk")
with tempfile.TemporaryDirectory() as d:
    argv, members, _ = write_partition_dir(d, b, parts, dim)
    for rep in range(3):
        if not (warm and rep == 2):
            shutil.rmtree(os.path.join(d, "models"), ignore_errors=True)
        shutil.rmtree(os.path.join(d, "ts"), ignore_errors=True)
        print("=== run", rep, flush=True)
        T0[0] = time.perf_counter()
        cli.run(argv)
        print(f"   total {(time.perf_counter()-T0[0])*1e3:.1f} ms", flush=True)
