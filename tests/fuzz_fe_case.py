"""The seeded random fixed-effect shards of tools/fuzz_fe.py, as a function: the sweep, the tool that takes one case apart on the
CPU (tools/fe_case_cpu.py) and the named regression tests (tests/test_fixed_effect.py) draw the same case from a seed."""
from types import SimpleNamespace

import numpy as np


def draw(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([50, 3000, 40000, 300000, 700000]))
    n = int(n * (0.5 + rng.random()))
    D = int(rng.choice([5, 300, 20000, 150000]))
    kmax = int(rng.choice([1, 4, 12, 40]))
    k = rng.integers(0, kmax + 1, n)
    if rng.random() < 0.3:    # a few very long rows
        k[rng.integers(0, n, 3)] = rng.integers(3000, 9000, 3)
    rp = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
    Z = int(rp[-1])
    cols = rng.integers(0, D, Z)
    if rng.random() < 0.4 and Z:   # dominant columns
        hot = rng.random(Z) < 0.3
        cols[hot] = rng.integers(0, min(D, 3), int(hot.sum()))
    vals = (rng.standard_normal(Z) * float(rng.choice([0.1, 1.0]))).astype(np.float32)
    linear = bool(rng.random() < 0.35)
    ic = bool(rng.random() < 0.8)
    w_true = rng.standard_normal(D) * 0.3
    z = np.zeros(n)
    np.add.at(z, np.repeat(np.arange(n), k), vals.astype(np.float64) * w_true[cols])
    off = (0.2 * rng.standard_normal(n)).astype(np.float32) if rng.random() < 0.7 else None
    wt = (0.5 + rng.random(n)).astype(np.float32) if rng.random() < 0.5 else None
    y = (z + 0.1 * rng.standard_normal(n)).astype(np.float32) if linear else (rng.random(n) < 1 / (1 + np.exp(-z))).astype(np.float32)
    l2 = float(rng.choice([0.1, 1.0, 10.0, 100.0]))
    regb = bool(rng.random() < 0.5)
    max_iter = int(rng.choice([3, 30, 200]))
    m = int(rng.choice([3, 10]))
    th0 = 0.05 * rng.standard_normal(D + (1 if ic else 0)) if rng.random() < 0.3 else None
    if th0 is not None:   # the oracle works in the space of the features present in the shard: start the absent ones at 0
        absent = np.ones(D, bool)
        absent[cols] = False
        th0[:D][absent] = 0.0
    return SimpleNamespace(n=n, D=D, Z=Z, k=k, rp=rp, cols=cols, vals=vals, y=y, off=off, wt=wt, linear=linear, ic=ic, l2=l2, regb=regb,
                           max_iter=max_iter, m=m, th0=th0)
