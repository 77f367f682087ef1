import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import *
from gdmix_amd.solver import REDeviceSolver, SolverOptions
s = REDeviceSolver(0)
b, opts, exp, _ = load_fixture("exit_extreme_02")
kw = opts_kwargs(opts)
packed = s.pack(b, has_intercept=True)
res = s.solve(packed, SolverOptions(**kw)).to_host()
print("e5", res["nit"][5], res["nfev"][5], res["status"][5], "ref", exp["nit"][5], exp["nfev"][5])
