import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import *
from gdmix_amd.solver import REDeviceSolver, SolverOptions
from oracle import oracle
s = REDeviceSolver(0)
for name in [n for n in fixture_names() if n.startswith("exit_")]:
    b, opts, exp, _ = load_fixture(name)
    kw = opts_kwargs(opts)
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    ref = oracle.solve(pk, b.val, b.y, b.offset, b.weight, oracle.make_opts(**kw))
    for mask, lds in ((7, 65536), (1, 65536), (2, 65536), (7, 0)):
        s.set_kernel_mask(mask); s.set_wave_lds_limit(lds)
        packed = s.pack(b, has_intercept=kw["has_intercept"])
        res = s.solve(packed, SolverOptions(**kw)).to_host()
        st = exp["strict"].astype(bool)
        cp = packed.coef_ptr_host()
        err = per_entity_rel_err(res["theta"], exp["theta"], cp)
        bad = st & ((res["nit"] != exp["nit"]) | (res["nfev"] != ref["neval"]) | (res["status"] != exp["status"]) | (err > 1e-7))
        print(name, "mask", mask, "lds", lds, "strict", st.sum(), "bad", bad.sum())
        for e in np.flatnonzero(bad)[:4]:
            print("   e", e, "dev nit/nfev/st", res["nit"][e], res["nfev"][e], res["status"][e], "oracle", ref["nit"][e], ref["neval"][e], ref["nfev"][e], ref["status"][e],
                  "ref", exp["nit"][e], exp["nfev"][e], exp["status"][e], "err %.2e" % err[e], "f dev %.17g or %.17g" % (res["fval"][e], ref["fval"][e]))
