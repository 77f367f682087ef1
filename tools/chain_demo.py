#!/usr/bin/env python3
"""One pass of GDMix's coordinate descent on one MI355X through the drop-in CLI (gdmix_amd/chain.py): global fixed effect ->
per-user -> per-movie random effect on MovieLens-100K-shaped data with planted effects; per-stage wall time and AUC.

    PYTHONPATH=. python tools/chain_demo.py [users] [movies] [ratings] [--child] [--keep DIR]
"""
import json
import sys
import tempfile

from gdmix_amd import chain

args = [a for a in sys.argv[1:] if not a.startswith("--")]
users = int(args[0]) if len(args) > 0 else chain.USERS
movies = int(args[1]) if len(args) > 1 else chain.MOVIES
ratings = int(args[2]) if len(args) > 2 else chain.RATINGS
child = "--child" in sys.argv
keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
data = chain.make_dataset(users, movies, ratings)
print(f"{ratings} ratings of {users} users x {movies} movies, {int(data['train'].sum())} training samples; AUC of the planted logit "
      f"{chain.auc(data['response'], data['true_logit']):.4f}", flush=True)
with tempfile.TemporaryDirectory() as d:
    for rep in ("first pass (HIP context, libraries)", "second pass"):
        import shutil
        root = keep or d
        shutil.rmtree(root, ignore_errors=True)
        res = chain.run_chain(root, data, child_process=child, log=lambda s: print("   ", s, flush=True))
        print(f"{rep}: {res['total_s']:.2f} s for the chain ({'child processes' if child else 'in process'})")
print(json.dumps(res))
