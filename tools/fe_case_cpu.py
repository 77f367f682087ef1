"""One fixed-effect fuzz case taken apart on the CPU (no device; scipy is the reference's optimiser, oracle/ is ours):

    PYTHONPATH=.:tests python tools/fe_case_cpu.py <seed> [iterations] [m]

1. scipy's fmin_l_bfgs_b on the case's objective (numpy; what fixed_effect_lr_lbfgs_model.py:635 calls with TensorFlow's value and
   gradient) against the oracle: status, iterations, evaluations, the objective after every iteration.
2. L-BFGS with L-BFGS-B's rules (lnsrlb's dcsrch, matupd's curvature test, the direction re-derived as (x + d) - x) in numpy, the
   direction by (a) the two-loop recursion, (b) the compact form with S'g and Y'g kept as running sums of the products with y — what
   the device kernels did in rounds 3 - 5 —, (c) the compact form with S'g and Y'g taken directly: every iterate against scipy's.
Round 6 (case 6700230): (b) leaves scipy's trajectory 300 times further than (a) and (c) at the first iterate after the oldest pair is
dropped, and the next iteration amplifies every perturbation 10^4 times; profiles/r06_fuzz.txt."""
import sys

import numpy as np
import scipy.sparse as sp
from scipy.optimize import fmin_l_bfgs_b
from scipy.optimize._dcsrch import DCSRCH

from gdmix_amd import fixed_effect as fe
from oracle import oracle
import fuzz_fe_case

seed = int(sys.argv[1])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
c = fuzz_fe_case.draw(seed)
m = int(sys.argv[3]) if len(sys.argv) > 3 else c.m
n, D, ic, linear, l2 = c.n, c.D, c.ic, c.linear, c.l2
print(f"case {seed}: n={n} D={D} Z={c.Z} linear={linear} ic={ic} l2={l2} regb={c.regb} max_iter={c.max_iter} m={m} warm={c.th0 is not None} "
      f"off={c.off is not None} wt={c.wt is not None}")
if c.th0 is not None:
    raise SystemExit("warm-started cases are not wired into this tool")
batch, dummy = fe.shard_as_batch(c.rp, c.cols, c.vals, c.y, c.off, c.wt, ic, binary_labels=not linear)
pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
uq = np.asarray(pk["unique_global"])
nf = uq.size
P = nf + (1 if ic else 0)
# the objective in the oracle's local space: intercept FIRST, then the features present in the shard
loc = np.full(D, -1, np.int64)
loc[uq] = np.arange(nf)
X = sp.csr_matrix((c.vals.astype(np.float64), loc[c.cols], c.rp), shape=(n, nf))
offs = np.zeros(n) if c.off is None else c.off.astype(np.float64)
w = np.ones(n) if c.wt is None else c.wt.astype(np.float64)
yy = c.y.astype(np.float64)
i0 = 1 if ic else 0
reg = np.ones(P)
if ic and not c.regb:
    reg[0] = 0.0


def fg(th):
    zz = X @ th[i0:] + offs + (th[0] if ic else 0.0)
    if linear:      # squared_difference, not halved (fixed_effect_lr_lbfgs_model.py:356-358)
        r = zz - yy
        f = np.sum(w * r * r)
        gr = 2.0 * w * r
    else:
        f = np.sum(w * (np.logaddexp(0, zz) - yy * zz))
        gr = w * (1 / (1 + np.exp(-zz)) - yy)
    g = np.empty(P)
    g[i0:] = X.T @ gr
    if ic:
        g[0] = gr.sum()
    return f + 0.5 * l2 * np.sum(reg * th * th), g + l2 * reg * th


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


EPS = np.finfo(float).eps
xs = [np.zeros(P)]
x, f, info = fmin_l_bfgs_b(fg, np.zeros(P), m=m, factr=1e-12 / EPS, maxiter=c.max_iter, callback=lambda xk: xs.append(xk.copy()))
print(f"scipy  : {info['task']} nit {info['nit']} nfev {info['funcalls']} f {f:.12g}")
o = oracle.make_opts(l2=l2, regularize_bias=c.regb and ic, has_intercept=ic, m=m, max_iter=c.max_iter, threshold=0.0, sum_loss=True, linear=linear)
res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o)
print(f"oracle : status {int(res['status'][0])} nit {int(res['nit'][0])} nfev {int(res['nfev'][0])} f {res['fval'][0]:.12g}   theta rel err scipy vs oracle {rel(x, res['theta']):.2e}")


def two_loop(S, Y, g):
    q = g.copy()
    al = []
    for s, y in zip(S[::-1], Y[::-1]):
        a = (s @ q) / (y @ s)
        al.append(a)
        q -= a * y
    r = q * ((S[-1] @ Y[-1]) / (Y[-1] @ Y[-1])) if S else q
    for (s, y), a in zip(zip(S, Y), al[::-1]):
        r += s * (a - (y @ r) / (y @ s))
    return -r


def run(mode, iters):
    """mode: two | sums | direct"""
    x = np.zeros(P)
    f, g = fg(x)
    S, Y = [], []
    ap = bp = np.zeros(0)
    SY = YY = np.zeros((0, 0))
    theta = 1.0
    out = []
    for it in range(iters):
        if mode == "two" or not S:
            d = two_loop(S, Y, g)
        else:
            a, b = (np.array(S) @ g, np.array(Y) @ g) if mode == "direct" else (ap, bp)
            R, gamma = np.triu(SY), 1.0 / theta
            q = np.linalg.solve(R, a)
            u = np.linalg.solve(R.T, (np.diag(np.diag(SY)) + gamma * YY) @ q - gamma * b)
            d = gamma * (np.array(Y).T @ q - g) - np.array(S).T @ u
        d = (x + d) - x
        gd0 = g @ d
        cache = {}

        def ev(t, x=x, d=d, f=f, g=g, cache=cache):
            if t not in cache:
                cache[t] = fg(x + t * d) if t != 0.0 else (f, g)
            return cache[t]
        ls = DCSRCH(lambda t: ev(t)[0], lambda t: ev(t)[1] @ d, 1e-3, 0.9, 0.1, 0.0, 1e10)
        stp, _, _, task = ls(min(1.0 / np.sqrt(g @ g), 1e10) if it == 0 else 1.0, phi0=f, derphi0=gd0, maxiter=20)
        if stp is None:
            break
        fn, gn = ev(stp)
        gd1 = gn @ d
        y, s = gn - g, stp * d
        dr, ddum = (gd1 - gd0) * stp, -gd0 * stp
        if dr > EPS * ddum:
            sy = np.array([si @ y for si in S])
            yyv = np.array([yi @ y for yi in Y])
            if len(S) == m:      # the oldest pair goes
                S.pop(0); Y.pop(0)
                SY, YY, ap, bp, sy, yyv = SY[1:, 1:], YY[1:, 1:], ap[1:], bp[1:], sy[1:], yyv[1:]
            k = len(S)
            ap, bp = np.append(ap + sy, stp * gd1), np.append(bp + yyv, y @ gn)      # the running sums
            SY2 = np.zeros((k + 1, k + 1)); SY2[:k, :k] = SY; SY2[:k, k] = sy; SY2[k, k] = dr
            YY2 = np.zeros((k + 1, k + 1)); YY2[:k, :k] = YY; YY2[:k, k] = yyv; YY2[k, :k] = yyv; YY2[k, k] = y @ y
            SY, YY = SY2, YY2
            S.append(s); Y.append(y)
            theta = (y @ y) / dr
        x, f, g = stp * d + x, fn, gn
        out.append((x.copy(), np.sqrt(g @ g)))
    return out


K = min(K, len(xs) - 1)
A, B, C = run("two", K), run("sums", K), run("direct", K)
print("iterate against scipy's (theta rel err):")
print(f"{'iter':>4s} {'|g|':>10s} {'two-loop':>10s} {'compact, sums':>14s} {'compact, direct':>16s}")
for k in range(min(K, len(A), len(B), len(C))):
    print(f"{k + 1:4d} {A[k][1]:10.3e} {rel(A[k][0], xs[k + 1]):10.2e} {rel(B[k][0], xs[k + 1]):14.2e} {rel(C[k][0], xs[k + 1]):16.2e}")
