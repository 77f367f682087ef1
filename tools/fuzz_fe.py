"""Randomised parity sweep of the fixed-effect solver: stepping kernels (include/gdmix_fe.h) and the device-wide team
kernel against the CPU oracle on seeded random shards. A tool for the GPU box, not a test.

    PYTHONPATH=.:tests python tools/fuzz_fe.py [cases] [first_seed]
"""
import os
import sys
import time

import numpy as np

from gdmix_amd import fixed_effect as fe
from gdmix_amd.solver import REDeviceSolver
from oracle import oracle
import fuzz_fe_case   # tests/fuzz_fe_case.py: the draws

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = REDeviceSolver(0)
s = fe.FixedEffectDeviceSolver(solver=dev)


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


bad = 0
t0 = time.time()
worst = 0.0
for case in range(cases):
    c = fuzz_fe_case.draw(seed0 + case)
    n, D, Z, rp, cols, vals, y, off, wt = c.n, c.D, c.Z, c.rp, c.cols, c.vals, c.y, c.off, c.wt
    linear, ic, l2, regb, max_iter, m, th0 = c.linear, c.ic, c.l2, c.regb, c.max_iter, c.m, c.th0
    for kv in filter(None, os.environ.get("FUZZ_FE_OVERRIDE", "").split(",")):   # one case taken apart: FUZZ_FE_OVERRIDE=m=3,l2=1.0 (after every draw, so the data stay the seed's)
        key, v = kv.split("=")
        if key == "m": m = int(v)
        elif key == "l2": l2 = float(v)
        elif key == "max_iter": max_iter = int(v)
        else: raise SystemExit(f"FUZZ_FE_OVERRIDE: unknown key {key}")
    mt = fe.LINEAR_REGRESSION if linear else fe.LOGISTIC_REGRESSION
    kw = dict(offset=off, weight=wt, has_intercept=ic, l2=l2, regularize_bias=regb, model_type=mt, theta0=th0, max_iter=max_iter, m=m)
    th_step, info_step = s.fit_stepping(rp, cols, vals, y, D, **kw)
    th_team, info_team = s.fit(rp, cols, vals, y, D, **kw) if Z < 4_000_000 else (None, None)
    batch, dummy = fe.shard_as_batch(rp, cols, vals, y, off, wt, ic, binary_labels=not linear)
    pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
    Dg = 1 if dummy else D
    o = oracle.make_opts(l2=l2, regularize_bias=regb and ic, has_intercept=ic, m=m, max_iter=max_iter, threshold=0.0, sum_loss=True, linear=linear)
    t0l = None
    if th0 is not None:
        t0l = fe.to_local(th0 if not dummy else np.concatenate([np.zeros(Dg), th0[D:]]), pk["unique_global"], Dg, ic, False)
    res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o, theta0=t0l)
    th_o = fe.to_global(res["theta"], pk["unique_global"], Dg, ic, False)
    if dummy:
        th_o = th_o[Dg:]
    # How reproducible is the oracle's own answer? Long runs amplify rounding (see tools/fuzz_parity.py): the same solve
    # from a start moved by 1e-15 .. 1e-13.
    sens, stable = 0.0, True
    P_loc = int(res["theta"].size)
    for j, mag in enumerate((1e-15, 1e-14, 1e-13)):
        jig = mag * np.random.default_rng(j + 1).standard_normal(P_loc)
        pert = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o, theta0=jig if t0l is None else t0l * (1.0 + jig))
        sens = max(sens, rel_err(pert["theta"], res["theta"]))
        stable = stable and int(pert["nit"][0]) == int(res["nit"][0]) and int(pert["status"][0]) == int(res["status"][0])
    # ... and from the same start with rounding in EVERY evaluation: the oracle with plain fp64 running sums instead of its long-double
    # accumulators (ORACLE_FE_NARROW) — what a kernel with another summation order applies; a start perturbation is damped by the
    # first iterations, this one is not (profiles/r04_fuzz.txt, cases 1100036 and 1100226)
    os.environ["ORACLE_FE_NARROW"] = "1"
    try:
        pert = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o, theta0=t0l)
    finally:
        os.environ.pop("ORACLE_FE_NARROW", None)
    sens = max(sens, rel_err(pert["theta"], res["theta"]))
    stable = stable and int(pert["nit"][0]) == int(res["nit"][0]) and int(pert["status"][0]) == int(res["status"][0])
    stable = stable and sens < 1e-9
    if os.environ.get("FUZZ_FE_DETAIL"):
        # one case in detail (FUZZ_FE_DETAIL=1 python tools/fuzz_fe.py 1 <seed>): objective values, counts, and the oracle under a wider
        # set of start perturbations — is the disagreement larger than what the reference's own arithmetic does to itself?
        print(f"oracle           status {int(res['status'][0])} nit {int(res['nit'][0])} nfev {int(res['nfev'][0])} f {res['fval'][0]:.12g} |g| {res['gnorm'][0]:.3e}")
        for name, th, info in (("stepping", th_step, info_step), ("team", th_team, info_team)):
            if th is not None:
                print(f"device {name:9s} status {int(info['status'])} nit {int(info['nit'])} nfev {int(info['nfev'])} f {float(info['fval']):.12g} |g| {float(info['gnorm']):.3e} "
                      f"theta rel err vs oracle {rel_err(th, th_o):.2e}")
        for j, mag in enumerate((1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-13, 1e-12, 1e-11, 1e-10)):
            jig = mag * np.random.default_rng(100 + j).standard_normal(P_loc)
            pert = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o, theta0=jig if t0l is None else t0l * (1.0 + jig))
            print(f"oracle, start noise {mag:.0e}: status {int(pert['status'][0])} nit {int(pert['nit'][0])} nfev {int(pert['nfev'][0])} f {pert['fval'][0]:.12g} "
                  f"theta rel err vs oracle {rel_err(pert['theta'], res['theta']):.2e}")
    if os.environ.get("FUZZ_FE_DETAIL"):
        # ... and the oracle with plain fp64 running sums instead of its long-double accumulators (ORACLE_FE_NARROW): rounding in EVERY
        # evaluation, which is what a kernel with another summation order applies — a start perturbation is damped by the first iterations
        os.environ["ORACLE_FE_NARROW"] = "1"
        try:
            nar = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o, theta0=t0l)
        finally:
            os.environ.pop("ORACLE_FE_NARROW", None)
        print(f"oracle, fp64 running sums: status {int(nar['status'][0])} nit {int(nar['nit'][0])} nfev {int(nar['nfev'][0])} f {nar['fval'][0]:.12g} "
              f"theta rel err vs oracle {rel_err(nar['theta'], res['theta']):.2e}")
    if os.environ.get("FUZZ_FE_ITERS"):
        # iteration by iteration: the same fit cut off after k iterations, device (stepping kernels) against the oracle — do the two
        # trajectories part gradually (rounding amplified by the problem) or at once (a defect)?
        for kk in [int(x) for x in os.environ["FUZZ_FE_ITERS"].split(",")]:
            kw_k = dict(kw, max_iter=kk)
            th_k, info_k = s.fit_stepping(rp, cols, vals, y, D, **kw_k)
            o_k = oracle.make_opts(l2=l2, regularize_bias=regb and ic, has_intercept=ic, m=m, max_iter=kk, threshold=0.0, sum_loss=True, linear=linear)
            r_k = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o_k, theta0=t0l)
            jig = 1e-13 * np.random.default_rng(7).standard_normal(P_loc)
            r_j = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o_k, theta0=jig if t0l is None else t0l * (1.0 + jig))
            os.environ["ORACLE_FE_NARROW"] = "1"
            try:
                r_n = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o_k, theta0=t0l)
            finally:
                os.environ.pop("ORACLE_FE_NARROW", None)
            tho_k = fe.to_global(r_k["theta"], pk["unique_global"], Dg, ic, False)
            print(f"max_iter {kk:3d}: oracle nit {int(r_k['nit'][0])} nfev {int(r_k['nfev'][0])} f {r_k['fval'][0]:.12g} | device nit {int(info_k['nit'])} nfev {int(info_k['nfev'])} "
                  f"f {float(info_k['fval']):.12g} theta rel err {rel_err(th_k, tho_k):.2e} | oracle + 1e-13 start noise: {rel_err(r_j['theta'], r_k['theta']):.2e} | "
                  f"oracle with fp64 running sums: nfev {int(r_n['nfev'][0])} f {r_n['fval'][0]:.12g} theta rel err {rel_err(r_n['theta'], r_k['theta']):.2e}")
    problems = []
    for name, th, info in (("stepping", th_step, info_step), ("team", th_team, info_team)):
        if th is None:
            continue
        e = rel_err(th, th_o)
        st = int(info["status"])
        if stable:
            tol = 1e-5 if (st == 1 or res["status"][0] == 1) else 1e-7
            if st != int(res["status"][0]) and not (st in (0, 1) and res["status"][0] in (0, 1)):
                problems.append(f"{name}: status {st} vs oracle {int(res['status'][0])}")
            # (a stop test met within rounding of its threshold may fall on the next iteration: then the coefficients agree)
            if st not in (1,) and res["status"][0] != 1 and int(info["nit"]) != int(res["nit"][0]) and e > 1e-9:
                problems.append(f"{name}: nit {info['nit']} vs oracle {int(res['nit'][0])}")
            worst = max(worst, e)
        else:
            tol = max(1e-6, 1000.0 * sens)
        if e > tol:
            problems.append(f"{name}: theta rel err {e:.2e} (status {st}, nit {info['nit']} / oracle {int(res['nit'][0])}; "
                            f"oracle sensitivity {sens:.1e}, {'reproducible' if stable else 'rounding-sensitive'})")
    if problems:
        bad += 1
    if problems or case % 5 == 0:
        print(f"{'BAD' if problems else 'ok '} case {seed0 + case} n={n} D={D} Z={Z} linear={linear} ic={ic} l2={l2} regb={regb} max_iter={max_iter} m={m} "
              f"warm={th0 is not None} off={off is not None} wt={wt is not None} dummy={dummy} {'reproducible' if stable else 'sensitive %.0e' % sens}" + "".join("\n      " + p for p in problems), flush=True)
print(f"{cases} cases, {bad} with disagreements, worst theta rel err on reproducible cases {worst:.2e}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
