#!/usr/bin/env python3
"""Generate tests/golden/fe_*.npz by RUNNING THE REFERENCE's own fixed-effect ground truth in the build container.

The reference's fixed-effect model computes its objective with TensorFlow ops (not installable here); what its test
suite compares that model against is a numpy + scipy statement of the same objective,
    gdmix-trainer/test/models/custom/test_fixed_effect_lr_lbfgs_model.py: _solve_for_coefficients (:480-527),
    _create_expected_data (:379-464), _predict (:467-477),
i.e. sum_i loss_i + (l2/2)|theta|^2 over [features | 1] with the intercept LAST, minimised by
scipy.optimize.fmin_l_bfgs_b(m=10, factr=1e-12, maxiter). This script imports that test module from
/root/reference (tensorflow, fastavro, smart_arg, statsmodels replaced by permissive stubs: they are only touched
at import time on this path) and records inputs and expected outputs. Only data is written.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_fe_fixtures.py
"""
import collections
import collections.abc
import importlib.abc
import importlib.machinery
import os
import sys
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/gdmix-trainer/src")
sys.path.insert(0, "/root/reference/gdmix-trainer/test")
collections.Mapping = collections.abc.Mapping

STUBBED = ("tensorflow", "fastavro", "smart_arg", "statsmodels", "psutil", "detext")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUBBED:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__ = []
        m.__spec__ = spec
        if spec.name == "smart_arg":
            m.arg_suite = lambda cls: cls      # class decorator: must hand the class back
        if spec.name == "tensorflow":
            m.test.TestCase = object           # base class of the reference's test classes
        return m

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _StubFinder())

import models.custom.test_fixed_effect_lr_lbfgs_model as ref  # noqa: E402
from gdmix.util import constants  # noqa: E402
from scipy.optimize import fmin_l_bfgs_b  # noqa: E402


def one_case(name, seed, has_offset, has_intercept, intercept_only, model_type, use_previous_model, l2=1.0, max_iter=100,
             n=None, d=None):
    # the reference's own generator gives data and its expected coefficients for this seed
    if n is not None:
        ref._NUM_SAMPLES, ref._NUM_FEATURES = n, d
    both = ref._create_expected_data(has_offset, seed, use_previous_model, intercept_only, has_intercept, model_type)
    exp, vexp = both["training"], both["validation"]
    ref._NUM_SAMPLES, ref._NUM_FEATURES = 100, 10
    y = np.asarray(exp.labels, np.float64)
    off = np.asarray(exp.offsets, np.float32)
    if intercept_only:
        X32 = np.zeros((y.size, 0), np.float32)
    else:
        X32 = np.asarray(exp.features, np.float32)     # what the reference writes to its TFRecords (float32)
    cols = [X32.astype(np.float64)]
    if has_intercept or intercept_only:
        cols.append(np.ones((y.size, 1)))
    Xp = np.hstack(cols)
    # ground truth on exactly these float32-valued inputs, by the reference's own function
    theta0 = None
    if use_previous_model:
        theta0 = ref._solve_for_coefficients(Xp, y, off.astype(np.float64), 100, model_type=model_type, l2_reg_weight=l2)
    theta = ref._solve_for_coefficients(Xp, y, off.astype(np.float64), max_iter if not use_previous_model else 1, theta0,
                                        model_type=model_type, l2_reg_weight=l2)
    rows, colsi = np.nonzero(X32)
    order = np.lexsort((colsi, rows))
    rows, colsi = rows[order], colsi[order]
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=y.size))]).astype(np.int64)
    # validation set of the same seed and the scores the reference's _predict gives for theta on float32-valued inputs
    voff = np.asarray(vexp.offsets, np.float32)
    VX32 = np.zeros((voff.size, 0), np.float32) if intercept_only else np.asarray(vexp.features, np.float32)
    vcols = [VX32.astype(np.float64)]
    if has_intercept or intercept_only:
        vcols.append(np.ones((voff.size, 1)))
    per_t, tot_t = ref._predict(theta, Xp, off.astype(np.float64))
    per_v, tot_v = ref._predict(theta, np.hstack(vcols), voff.astype(np.float64))
    vr, vc = np.nonzero(VX32)
    vo = np.lexsort((vc, vr))
    vr, vc = vr[vo], vc[vo]
    vrp = np.concatenate([[0], np.cumsum(np.bincount(vr, minlength=voff.size))]).astype(np.int64)
    val_out = dict(v_row_nnz_ptr=vrp, v_col_global=vc.astype(np.int64), v_val=VX32[vr, vc].astype(np.float32),
                   v_y=np.asarray(vexp.labels, np.float32), v_offset=voff, train_per_coord=per_t, train_score=tot_t,
                   valid_per_coord=per_v, valid_score=tot_v)
    out = dict(**val_out, row_nnz_ptr=row_ptr, col_global=colsi.astype(np.int64), val=X32[rows, colsi].astype(np.float32),
               y=y.astype(np.float32), offset=off, num_features=np.int64(X32.shape[1]),
               has_intercept=np.int64(1 if (has_intercept or intercept_only) else 0), l2=np.float64(l2),
               linear=np.int64(model_type == constants.LINEAR_REGRESSION), max_iter=np.int64(max_iter if not use_previous_model else 1),
               theta0=np.zeros(0) if theta0 is None else theta0, theta=theta,
               ref_expected_f32=np.asarray(exp.coefficients, np.float32))
    np.savez_compressed(os.path.join(HERE, f"fe_{name}.npz"), **out)
    print(f"fe_{name}: n={y.size} d={X32.shape[1]} theta[:4]={theta[:4]}")


LOGIT, LIN = constants.LOGISTIC_REGRESSION, constants.LINEAR_REGRESSION
one_case("logistic_offset", 1, True, True, False, LOGIT, False)
one_case("logistic_no_offset", 2, False, True, False, LOGIT, False)
one_case("logistic_no_intercept", 3, True, False, False, LOGIT, False)
one_case("logistic_intercept_only", 4, True, True, True, LOGIT, False)
one_case("logistic_warm_one_iteration", 5, True, True, False, LOGIT, True)
one_case("linear_offset", 6, True, True, False, LIN, False)
one_case("linear_no_intercept", 7, False, False, False, LIN, False)
one_case("logistic_l2_0.01", 8, True, True, False, LOGIT, False, l2=0.01)
one_case("logistic_3_iterations", 9, True, True, False, LOGIT, False, max_iter=3)
one_case("logistic_wide", 10, True, True, False, LOGIT, False, n=400, d=300)
one_case("linear_wide", 11, True, True, False, LIN, False, n=400, d=300)
