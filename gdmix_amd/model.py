"""RandomEffectLRLBFGSModel on MI355X: the reference's RE model API
(gdmix-trainer/src/gdmix/models/custom/random_effect_lr_lbfgs_model.py:56-317, models/api.py:4-84) with the
producer/consumer pool of scipy solves replaced by one batched device call per partition.

What stays as in the reference: parameters (REParams), directory conventions (active/passive), warm start
from `output_model_dir/part-000KK.avro`, carrying over prior entities that are absent from the new data,
the photon-ml Avro model format, the score Avro format, scoring of validation / active / passive data
after training. What changes: per_entity_grouped_input_fn + prepare_jobs + N scipy consumers become
read_grouped_partition -> gdmix_re_pack -> gdmix_re_solve / gdmix_re_score (gdmix_amd/csrc).
"""
import logging
import os
import struct
import threading
from collections import namedtuple

import numpy as np

from . import constants
from .io import avro, native_reader
from .io.features import get_feature_map, read_feature_list
from .io.grouped_reader import read_grouped_partition
from .io.metadata import DatasetMetadata, read_json_file
from .params import REParams
from .batch import WireRawBatch
from .solver import REDeviceSolver, SolverOptions, VARIANCE_MODES, host_array

logger = logging.getLogger(__name__)
logger.setLevel(logging.INFO)

# entity_id -> TrainingResult, exactly the reference's value type (scipy/job_consumers.py:18)
TrainingResult = namedtuple("TrainingResult", ("theta", "variance", "unique_global_indices"))


def torch_int64():
    """torch.int64 without importing torch at module import (the host tests run without a device). The device keeps a batch's local ->
    global feature map as int32 (ABI 12); the host side of this file — model table, Avro writer, coefficient mapping — works on int64:
    widened on the device before the copy (a 10 us kernel) rather than by numpy on the host (a few ms per partition)."""
    import torch
    return torch.int64


_ROW_BITS = 40
_ROW_MASK = (1 << _ROW_BITS) - 1
WRITE_BEHIND_THREADS = int(os.environ.get("GDMIX_WRITE_THREADS", "8"))   # files being written at a time behind the device work
# Consecutive cold-start partitions solved in ONE device batch (plan_group): up to GROUP_MAX of them and GROUP_MAX_BYTES of input files
# (GDMIX_PARTITIONS_PER_BATCH=1: every partition on its own, as the reference trains them). A partition of a hashed population is a small
# batch for an MI355X — one rank's share of BASELINE config 5, one partition per launch, 3.8 M entities/s on the device; two / four / eight
# per launch 4.6 / 5.1 / 5.3 M (profiles/r05_c5_full_share.txt): what a partition waits for is the latency of its few largest entities,
# which a larger batch shares. Through the CLI the group also has to wait for all its partitions to be decoded and hands all their files
# to the writers at once, so small groups win: a C5-shaped run of 8 partitions of 63 MB, 0.91 - 0.96 M entities/s one by one, 1.07 - 1.09 M
# in pairs, 0.94 - 0.96 in threes, 1.01 - 1.03 in fours, 0.84 - 0.88 all eight at once; C2's 150 MB partitions lose a tenth when paired
# (their device time is already amortised, their files are what the run waits for). Hence the byte bound: partitions of that size stay
# alone, smaller ones pair up (profiles/r05_group_ab.txt). The files that come out are per partition either way.
GROUP_MAX = max(1, int(os.environ.get("GDMIX_PARTITIONS_PER_BATCH", "4")))
GROUP_MAX_BYTES = int(float(os.environ.get("GDMIX_GROUP_MB", "160")) * (1 << 20))


class ModelTable:
    """entity id -> TrainingResult, stored as flat arrays per chunk so that a million models do not become
    a million Python objects. Iteration order follows dict.update semantics: existing ids keep their place,
    new ids are appended (random_effect_lr_lbfgs_model.py:162)."""

    def __init__(self):
        self._chunks = []      # dict(theta, variance|None, idx, coef_ptr, feat_ptr)
        self._where = {}       # id -> chunk index << 40 | row
        # Chunks whose ids are not in _where yet: [(chunk index, ids)]. The dict is only built when somebody looks an id up
        # (a warm start, an inference run, a carry-over); a cold training run adds a partition's models and exports them, and its
        # ids — all different, checked natively on the bytes — never become dict entries (10 ms of interpreter time per 125 k).
        self._pending = []
        self._plan_cache = None
        # the table of a partition is exported by a writer thread while the main thread scores with it: the lazy steps below
        # (building the dict, planning the merged order) happen under a lock whichever thread needs them first
        self._lock = threading.RLock()

    def _index(self):
        with self._lock:
            self._plan_cache = None
            pending, self._pending = self._pending, []
            for c, ids in pending:
                base = c << _ROW_BITS
                if not self._where:
                    self._where = dict(zip(ids, range(base, base + len(ids))))
                else:
                    self._where.update(zip(ids, range(base, base + len(ids))))   # an existing key keeps its position, only the value changes
            return self._where

    def _segments(self):
        """[(chunk, ids)] when the table is still nothing but its chunks in order — not indexed, every chunk's ids native
        (EntityIds) and all different —, else None. Such a table answers through native joins of the id bytes."""
        if self._where or not self._pending or len(self._pending) != len(self._chunks):
            return None
        for k, (c, ids) in enumerate(self._pending):
            if c != k or not hasattr(ids, "all_different") or not ids.all_different():
                return None
        return self._pending

    def _plan(self):
        """(ids, chunk-and-row per id) of the table in dict.update order without building the dict: a prior model and the
        models trained on top of it (existing ids keep their place and take the new model, new ids are appended). One or two
        chunks; None otherwise."""
        with self._lock:
            return self._plan_locked()

    def _plan_locked(self):
        if self._plan_cache is not None:
            return self._plan_cache
        seg = self._segments()
        if seg is None or len(seg) > 2:
            return None
        ids0 = seg[0][1]
        n0 = len(ids0)
        cr = np.zeros((n0, 2), np.int64)
        cr[:, 1] = np.arange(n0)
        ids = ids0
        if len(seg) == 2:
            ids1 = seg[1][1]
            at = ids1.rows_in(ids0)                  # row of the prior table, or -1
            hit = np.flatnonzero(at >= 0)
            cr[at[hit], 0] = 1
            cr[at[hit], 1] = hit
            new = np.flatnonzero(at < 0)
            if new.size:
                tail = np.ones((new.size, 2), np.int64)
                tail[:, 1] = new
                cr = np.concatenate([cr, tail])
                ids = ids0.extended(ids1.take(new))
                ids._unique = True
        self._plan_cache = (ids, cr)
        return self._plan_cache

    def __len__(self):
        plan = self._plan()
        return len(plan[0]) if plan is not None else len(self._index())

    def __bool__(self):
        return bool(self._where) or any(len(ids) for _, ids in self._pending)

    def __contains__(self, k):
        return k in self._index()

    def add_chunk(self, ids, theta, coef_ptr, idx, feat_ptr, variance=None):
        c = len(self._chunks)
        self._chunks.append(dict(theta=np.asarray(theta, np.float64), variance=None if variance is None else np.asarray(variance, np.float64),
                                 idx=np.asarray(idx, np.int64), coef_ptr=np.asarray(coef_ptr, np.int64),
                                 feat_ptr=np.asarray(feat_ptr, np.int64)))
        self._pending.append((c, ids))
        self._plan_cache = None

    def get(self, k, default=None):
        w = self._index().get(k)
        if w is None:
            return default
        c, r = w >> _ROW_BITS, w & _ROW_MASK
        ch = self._chunks[c]
        a, b = ch["coef_ptr"][r], ch["coef_ptr"][r + 1]
        fa, fb = ch["feat_ptr"][r], ch["feat_ptr"][r + 1]
        return TrainingResult(ch["theta"][a:b], None if ch["variance"] is None else ch["variance"][a:b], ch["idx"][fa:fb])

    def __getitem__(self, k):
        v = self.get(k)
        if v is None:
            raise KeyError(k)
        return v

    def keys(self):
        return self._index().keys()

    def items(self):
        for k in self._index():
            yield k, self.get(k)

    def update(self, other):
        if isinstance(other, ModelTable):
            base = len(self._chunks)
            self._chunks.extend(other._chunks)
            self._plan_cache = None
            if not other._where:
                # the other table's chunks as they are, still unindexed: _index() / _plan() apply dict.update's rule when asked
                self._pending += [(base + c, ids) for c, ids in other._pending]
                return
            shift = base << _ROW_BITS
            self._index().update((k, w + shift) for k, w in other._index().items())
        else:
            for k, tr in dict(other).items():
                p = len(tr.theta)
                d = len(tr.unique_global_indices)
                self.add_chunk([k], tr.theta, [0, p], tr.unique_global_indices, [0, d], tr.variance)

    def flatten(self):
        """Flat arrays over the entities in dict order: (ids, coef_beg, coef_cnt, var_beg, feat_beg, mean, variance,
        feat_idx); var_beg is -1 for an entity whose chunk carries no variance, variance None if no chunk does."""
        plan = self._plan()
        if plan is not None and len(self._chunks) == 1:   # the chunk is the table: its arrays as they are
            ids = plan[0]
            ch = self._chunks[0]
            cp, fp = ch["coef_ptr"], ch["feat_ptr"]
            var_beg = cp[:-1].copy() if ch["variance"] is not None else np.full(len(ids), -1, np.int64)
            return ids, cp[:-1].copy(), np.diff(cp), var_beg, fp[:-1].copy(), ch["theta"], ch["variance"], ch["idx"]
        if plan is not None:
            ids, cr = plan
        else:
            self._index()
            ids = list(self._where.keys())
            w = np.fromiter(self._where.values(), np.int64, count=len(ids))
            cr = np.stack([w >> _ROW_BITS, w & _ROW_MASK], axis=1) if len(ids) else np.zeros((0, 2), np.int64)
        # Only what the table still points at goes into the flat arrays: a chunk none of whose models survived (a prior model
        # every entity of which was trained again) contributes nothing; a chunk used whole is taken as it is; of a chunk used in
        # part (the carried-over entities of a prior model) only those models are gathered.
        from .batch import _ranges
        E = len(ids)
        coef_beg = np.zeros(E, np.int64); coef_cnt = np.zeros(E, np.int64)
        feat_beg = np.zeros(E, np.int64); var_beg = np.full(E, -1, np.int64)
        means, idxs, variances = [], [], []
        co = fo = vo = 0
        for c, ch in enumerate(self._chunks):
            sel = np.flatnonzero(cr[:, 0] == c) if E else np.zeros(0, np.int64)
            if sel.size == 0:
                continue
            rows = cr[sel, 1]
            cp, fp = ch["coef_ptr"], ch["feat_ptr"]
            cnt = cp[rows + 1] - cp[rows]
            fcnt = fp[rows + 1] - fp[rows]
            whole = 2 * int(cnt.sum()) >= len(ch["theta"])
            if whole:
                cstart, fstart = cp[rows], fp[rows]
                means.append(ch["theta"]); idxs.append(ch["idx"])
                if ch["variance"] is not None:
                    variances.append(ch["variance"])
                n_c, n_f = len(ch["theta"]), len(ch["idx"])
            else:
                cstart = np.cumsum(cnt) - cnt
                fstart = np.cumsum(fcnt) - fcnt
                take_c = _ranges(cp[rows], cnt)
                means.append(ch["theta"][take_c]); idxs.append(ch["idx"][_ranges(fp[rows], fcnt)])
                if ch["variance"] is not None:
                    variances.append(ch["variance"][take_c])
                n_c, n_f = int(cnt.sum()), int(fcnt.sum())
            coef_beg[sel] = co + cstart
            coef_cnt[sel] = cnt
            feat_beg[sel] = fo + fstart
            if ch["variance"] is not None:
                var_beg[sel] = vo + cstart
                vo += n_c
            co += n_c; fo += n_f
        cat = lambda parts, dt: parts[0] if len(parts) == 1 else (np.concatenate(parts) if parts else np.zeros(0, dt))
        mean, idx = cat(means, np.float64), cat(idxs, np.int64)
        variance = cat(variances, np.float64) if variances else None
        return ids, coef_beg, coef_cnt, var_beg, feat_beg, mean, variance, idx

    def rows_for(self, ids):
        """The models of `ids` as flat arrays in that order: dict(has [E] bool, coef_ptr [E+1], theta, feat_ptr [E+1], idx);
        an id without a model has empty slices."""
        from .batch import _ranges
        E = len(ids)
        found, chunk, row = self.lookup(ids)
        cc = np.zeros(E, np.int64); fc = np.zeros(E, np.int64)
        cs = np.zeros(E, np.int64); fs = np.zeros(E, np.int64)
        co = fo = 0
        for c, ch in enumerate(self._chunks):
            sel = np.flatnonzero(found & (chunk == c))
            if sel.size:
                r = row[sel]
                cc[sel] = ch["coef_ptr"][r + 1] - ch["coef_ptr"][r]
                fc[sel] = ch["feat_ptr"][r + 1] - ch["feat_ptr"][r]
                cs[sel] = co + ch["coef_ptr"][r]
                fs[sel] = fo + ch["feat_ptr"][r]
            co += len(ch["theta"]); fo += len(ch["idx"])
        theta = np.concatenate([ch["theta"] for ch in self._chunks]) if self._chunks else np.zeros(0)
        idx = np.concatenate([ch["idx"] for ch in self._chunks]) if self._chunks else np.zeros(0, np.int64)
        return dict(has=found, coef_ptr=np.concatenate([[0], np.cumsum(cc)]).astype(np.int64), theta=theta[_ranges(cs, cc)],
                    feat_ptr=np.concatenate([[0], np.cumsum(fc)]).astype(np.int64), idx=idx[_ranges(fs, fc)])

    @classmethod
    def from_rows(cls, ids, rows):
        """Inverse of rows_for: a table of the ids that have a model."""
        t = cls()
        t._chunks.append(dict(theta=np.asarray(rows["theta"], np.float64), variance=None, idx=np.asarray(rows["idx"], np.int64),
                              coef_ptr=np.asarray(rows["coef_ptr"], np.int64), feat_ptr=np.asarray(rows["feat_ptr"], np.int64)))
        for r in np.flatnonzero(rows["has"]):
            t._where[ids[int(r)]] = int(r)
        return t

    def lookup(self, ids):
        """Vectorised: for every id, (found mask, chunk index, row)."""
        with self._lock:
            seg = list(self._segments() or []) if hasattr(ids, "rows_in") else None
        if seg:   # native joins, the later chunk's model wins
            row = np.full(len(ids), -1, np.int64)
            chunk = np.zeros(len(ids), np.int64)
            for c, seg_ids in reversed(seg):
                open_ = np.flatnonzero(row < 0) if c != seg[-1][0] else None
                at = ids.rows_in(seg_ids) if open_ is None else (ids.take(open_).rows_in(seg_ids) if open_.size else np.zeros(0, np.int64))
                if open_ is None:
                    row = at
                    chunk[:] = c
                else:
                    got = at >= 0
                    row[open_[got]] = at[got]
                    chunk[open_[got]] = c
            found = row >= 0
            return found, np.where(found, chunk, 0), np.where(found, row, 0)
        found = np.zeros(len(ids), bool)
        chunk = np.zeros(len(ids), np.int64)
        row = np.zeros(len(ids), np.int64)
        get = self._index().get
        w = np.fromiter((get(k, -1) for k in ids), np.int64, count=len(ids))
        found = w >= 0
        chunk = np.where(found, w >> _ROW_BITS, 0)
        row = np.where(found, w & _ROW_MASK, 0)
        return found, chunk, row


def _model_coefficients_for_batch(table, entity_ids, unique_global, ent_feat_ptr, has_intercept, num_features, native=None, out=None):
    """Coefficients of the prior / trained models in the packed batch's local index space.

    For every entity that has a model: intercept from the model (always), and for every feature present in
    the current data the model's coefficient if it has one, else 0 (prepare_jobs:262-288 for training warm
    start; InferenceJobConsumer for scoring). Returns (theta [P] float64, has_model [E] uint8)."""
    E = len(entity_ids)
    ic = 1 if has_intercept else 0
    d = np.diff(ent_feat_ptr)
    coef_ptr = ent_feat_ptr + np.arange(E + 1, dtype=np.int64) * ic
    if native is None:
        native = native_reader.available()
    cleared = True
    if out is not None:   # a caller-owned (page-locked) block of at least that size: cleared by the native mapping, in parallel
        theta = out[:int(coef_ptr[-1])]
        cleared = False
    else:
        theta = np.zeros(int(coef_ptr[-1]), np.float64)
    found, chunk, row = table.lookup(entity_ids)
    has_model = found.astype(np.uint8)
    if not found.any() or not native:
        if not cleared:
            theta[:] = 0.0
        if not found.any():
            return theta, has_model
    if native:
        for c in np.unique(chunk[found]):
            ch = table._chunks[int(c)]
            src_row = np.where(found & (chunk == c), row, -1)
            native_reader.map_coefficients(theta, ent_feat_ptr, unique_global, src_row, ch["coef_ptr"], ch["feat_ptr"], ch["theta"],
                                           ch["idx"], has_intercept, zero_first=not cleared)
            cleared = True
        return theta, has_model
    F = int(num_features) + 1
    ent_of_feat = np.repeat(np.arange(E, dtype=np.int64), d)
    cur_key = ent_of_feat * F + unique_global
    feat_coef_pos = coef_ptr[:-1][ent_of_feat] + ic + (np.arange(unique_global.size, dtype=np.int64) - ent_feat_ptr[:-1][ent_of_feat])
    for c in np.unique(chunk[found]):
        ch = table._chunks[int(c)]
        sel = np.flatnonzero(found & (chunk == c))
        rows = row[sel]
        if ic:
            theta[coef_ptr[sel]] = ch["theta"][ch["coef_ptr"][rows]]
        fa, fb = ch["feat_ptr"][rows], ch["feat_ptr"][rows + 1]
        cnt = fb - fa
        if cnt.sum() == 0:
            continue
        from .batch import _ranges
        src_feat = _ranges(fa, cnt)                              # positions in the chunk's idx array
        src_ent = np.repeat(sel, cnt)                            # batch entity of each prior feature
        within = src_feat - np.repeat(fa, cnt)
        src_val = ch["theta"][np.repeat(ch["coef_ptr"][rows], cnt) + ic + within]
        prior_key = src_ent * F + ch["idx"][src_feat]
        order = np.argsort(prior_key, kind="stable")
        pk, pv = prior_key[order], src_val[order]
        pos = np.searchsorted(pk, cur_key)
        pos_c = np.minimum(pos, pk.size - 1)
        hit = (pos < pk.size) & (pk[pos_c] == cur_key)
        theta[feat_coef_pos[hit]] = pv[pos_c[hit]]
    return theta, has_model


class RandomEffectLRLBFGSModel:
    """Per-entity L2-regularised logistic regression, all entities of a partition solved on one MI355X."""

    def __init__(self, raw_model_params, device=None):
        self.model_params: REParams = self._parse_parameters(raw_model_params)
        self.checkpoint_path = os.path.join(self.model_params.output_model_dir)
        self.metadata_file = self.model_params.metadata_file
        self.feature_bag_name = self.model_params.feature_bag
        self.has_intercept = self.model_params.has_intercept
        # If no features, then make sure feature file is None. This is intercept only model.
        self.feature_file = None if self.feature_bag_name is None else self.model_params.feature_file
        if self.model_params.training_data_dir is not None:
            self.training_data_dir = os.path.join(self.model_params.training_data_dir, constants.ACTIVE)
            self.passive_training_data_dir = os.path.join(self.model_params.training_data_dir, constants.PASSIVE)
        else:
            self.training_data_dir = None
            self.passive_training_data_dir = None
        self.validation_data_dir = self.model_params.validation_data_dir
        self.disable_random_effect_scoring_after_training = self.model_params.disable_random_effect_scoring_after_training
        self._device_index = device
        self._solver = None
        self._read_cache = None     # (key, batch) of the partition _train decoded last
        self._io_pool = None        # begin_pipeline(): files are read ahead and written behind the device work
        self._write_pool = None
        self._prefetched = {}       # read key -> Future[RawBatch]
        self._prefetched_models = {}   # model file -> Future[ModelTable]
        self._pending_writes = []
        self._group = None          # plan_group(): partitions that are solved in one device batch
        self._decoded = {}          # read key -> RawBatch decoded for a group, until _read hands it out
        self.last_training_stats = None

    # ---- Model API (models/api.py) ---------------------------------------------------------------------
    def train(self, training_data_dir, validation_data_dir, metadata_file, checkpoint_path, execution_context, schema_params):
        logger.info("Kicking off random effect custom LR training")
        self._action(constants.ACTION_TRAIN, (training_data_dir, validation_data_dir), metadata_file, checkpoint_path,
                     execution_context, schema_params)

    def predict(self, output_dir, input_data_path, metadata_file, checkpoint_path, execution_context, schema_params):
        logger.info(f"Running inference on dataset : {input_data_path}, results to be written to path : {output_dir}")
        self._action(constants.ACTION_INFERENCE, (output_dir, input_data_path), metadata_file, checkpoint_path,
                     execution_context, schema_params)

    def export(self, output_model_dir):
        logger.info("Model export is done as part of the training() API for random effect LR LBFGS training. Skipping.")

    def _parse_parameters(self, raw_model_parameters) -> REParams:
        params = REParams.__from_argv__(raw_model_parameters, error_on_unknown=False)
        logger.info(params)
        return params

    # ---- internals ----------------------------------------------------------------------------------
    def _get_solver(self):
        if self._solver is None:
            dev = self._device_index
            if dev is None:
                dev = int(os.environ.get("LOCAL_RANK", "0"))
                if os.environ.get("GDMIX_RANKS_SHARE_DEVICE") == "1":    # test hook for a 1-GPU box: several ranks on one device (gloo)
                    import torch
                    dev %= max(1, torch.cuda.device_count())
            self._solver = REDeviceSolver(dev)   # raises if no MI355X / no library: there is no CPU fallback
            if getattr(self.model_params, "rebalance_entities", False) and hasattr(self._solver, "pin_routing"):
                # an entity may be solved in another batch than the one its partition would have given it: its kernel must not
                # depend on the batch, or the same entity would come out with other last bits (REDeviceSolver.pin_routing)
                self._solver.pin_routing()
        return self._solver

    def _solver_options(self):
        mp = self.model_params
        # without an intercept the whole theta is regularised whatever regularize_bias says (binary_logistic_regression.py:72-82)
        return SolverOptions(l2=mp.l2_reg_weight, regularize_bias=bool(mp.regularize_bias) and bool(self.has_intercept),
                             has_intercept=self.has_intercept,
                             m=mp.num_of_lbfgs_curvature_pairs, max_iter=mp.num_of_lbfgs_iterations,
                             ftol=mp.lbfgs_tolerance, variance_mode=VARIANCE_MODES[mp.random_effect_variance_mode],
                             threshold=mp.sparsity_threshold)

    def _action(self, action, action_context, metadata_file, checkpoint_path, execution_context, schema_params):
        partition_index = execution_context[constants.PARTITION_INDEX]
        metadata = read_json_file(metadata_file)
        tensor_metadata = DatasetMetadata(metadata)
        # if intercept only model, pad a dummy feature, otherwise, read number of features from the metadata
        num_features = 1 if self.feature_bag_name is None else tensor_metadata.get_feature_shape(self.feature_bag_name)[0]
        logger.info(f"Found {num_features} features in feature bag {self.feature_bag_name}")
        assert num_features > 0, "number of features must > 0"
        avro_filename = f"part-{partition_index:05d}.avro"
        if action == constants.ACTION_INFERENCE:
            output_dir, input_data_path = action_context
            model_weights = self._load_weights(os.path.join(checkpoint_path, avro_filename))
            self._predict(input_path=input_data_path, tensor_metadata=tensor_metadata,
                          output_file=os.path.join(output_dir, avro_filename), model_weights=model_weights,
                          schema_params=schema_params, num_features=num_features)
        elif action == constants.ACTION_TRAIN:
            training_data_dir, validation_data_dir = action_context
            model_file = os.path.join(self.model_params.output_model_dir, avro_filename)
            model_weights = self._load_weights(model_file, True)    # load initial model if available
            model_weights = self._train(training_data_dir, tensor_metadata, model_weights, num_features, schema_params,
                                        model_file)

            def predict(input_path, output_file):
                self._predict(input_path=input_path, tensor_metadata=tensor_metadata, output_file=output_file,
                              model_weights=model_weights, schema_params=schema_params, num_features=num_features)
            if validation_data_dir:
                o = execution_context.get(constants.VALIDATION_OUTPUT_FILE, None)
                o and predict(validation_data_dir, o)
            if not self.disable_random_effect_scoring_after_training:
                o = execution_context.get(constants.ACTIVE_TRAINING_OUTPUT_FILE, None)
                o and predict(training_data_dir, o)
                i, o = execution_context.get(constants.PASSIVE_TRAINING_DATA_DIR, None), \
                    execution_context.get(constants.PASSIVE_TRAINING_OUTPUT_FILE, None)
                i and o and predict(i, o)
            self._read_cache = None
        else:
            raise ValueError(f"Invalid action {action!r}.")

    # ---- host pipeline: the driver reads partition k+1 and writes partition k's files while the device works -------
    def begin_pipeline(self):
        """From here on prefetch() decodes input in a background thread and the Avro files are written by one; flush()
        waits for them (and raises what they raised). The native reader / writers release the GIL."""
        if self._io_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            # readers and writers apart: a partition's two files take longer to write than the partition takes to solve, and a
            # read queued behind them would stall the device
            # Inside the pipeline three or four decodes and four to six Avro writers run at once: 32 threads per native call (the
            # library's default, the optimum of ONE call alone) put ~200 runnable threads on the box and every phase of every call
            # waited for the others (profiles/r04_host_path.txt: a decode of 11 ms alone took 45 - 75 ms). 12 per call, measured
            # 2 .. 64 (tools/r04_e2e_knobs.sh); the caller's GDMIX_IO_THREADS wins, and a box with fewer cores gets fewer. Held in
            # the reader module for the life of the pipeline (end_pipeline gives it back): the process environment is not touched.
            native_reader.pipeline_threads_begin(min(12, os.cpu_count() or 1))
            self._io_pool = ThreadPoolExecutor(max_workers=6, thread_name_prefix="gdmix-read")   # three partitions ahead + their prior models
            self._write_pool = ThreadPoolExecutor(max_workers=WRITE_BEHIND_THREADS, thread_name_prefix="gdmix-write")

    def _read_key(self, input_path, num_features):
        return (os.path.abspath(input_path), self.model_params.partition_entity, self.feature_bag_name, num_features)

    def prefetch(self, input_path, metadata_file, schema_params):
        """Start decoding the partition a later train() / predict() call will ask for."""
        if self._io_pool is None or not os.path.isdir(input_path):
            return
        tensor_metadata = DatasetMetadata(read_json_file(metadata_file))
        num_features = 1 if self.feature_bag_name is None else tensor_metadata.get_feature_shape(self.feature_bag_name)[0]
        key = self._read_key(input_path, num_features)
        if key not in self._prefetched and key not in self._decoded:
            self._prefetched[key] = self._io_pool.submit(self._read_ahead, input_path, tensor_metadata, schema_params, num_features)

    def prefetch_prior_model(self, partition_index):
        """Start loading the model file train() will warm-start partition `partition_index` from, if there is one."""
        model_file = os.path.abspath(os.path.join(self.model_params.output_model_dir, f"part-{partition_index:05d}.avro"))
        if self._io_pool is None or not os.path.exists(model_file) or model_file in self._prefetched_models:
            return
        self.flush(model_file)
        self._prefetched_models[model_file] = self._io_pool.submit(self._load_weights_from, model_file)

    def _write_behind(self, path, fn, *args, **kwargs):
        if self._io_pool is None:
            return fn(*args, **kwargs)
        # a write that already failed is raised now, before more partitions are trained on top of a missing file
        done = [(p, f) for p, f in self._pending_writes if f.done()]
        self._pending_writes = [(p, f) for p, f in self._pending_writes if not f.done()]
        for _, f in done:
            f.result()
        self._pending_writes.append((os.path.abspath(path), self._write_pool.submit(fn, *args, **kwargs)))

    def flush(self, path=None):
        """Wait for the files still being written (only `path` if given); the first failure is raised here."""
        path = None if path is None else os.path.abspath(path)
        wait = [(p, f) for p, f in self._pending_writes if path is None or p == path]
        self._pending_writes = [(p, f) for p, f in self._pending_writes if not (path is None or p == path)]
        for _, f in wait:
            f.result()

    def end_pipeline(self):
        try:
            self.flush()
        finally:
            for f in list(self._prefetched.values()) + list(self._prefetched_models.values()):
                f.cancel()
            self._prefetched, self._prefetched_models = {}, {}
            self._group, self._decoded = None, {}
            if self._io_pool is not None:
                self._io_pool.shutdown(wait=True)
                self._write_pool.shutdown(wait=True)
                self._io_pool = self._write_pool = None
                native_reader.pipeline_threads_end()
            self._read_cache = None
            # the pooled host blocks (up to GDMIX_IO_POOL_MB + 1 GB of writer buffers) go back to the allocator — behind the caller's
            # back: unmapping 5 GB of touched pages takes 36 - 105 ms (profiles/r04_host_path.txt), a fifth of a warm-started
            # million-entity run when it was done here in line
            import threading
            threading.Thread(target=native_reader.pool_trim, name="gdmix-pool-trim", daemon=True).start()

    def _read(self, input_path, tensor_metadata, schema_params, num_features, need_label):
        assert self.model_params.data_format == constants.TFRECORD
        # The active training data is scored right after it was trained on (_action): the partition is decoded once.
        key = self._read_key(input_path, num_features)
        if self._read_cache is not None and self._read_cache[0] == key:
            batch = self._read_cache[1]
            self._read_cache = None
            return batch
        batch = self._decoded.pop(key, None)
        ahead = self._prefetched.pop(key, None)
        if batch is None:
            batch = ahead.result() if ahead is not None else self._read_files(input_path, tensor_metadata, schema_params, num_features)
        if need_label:
            self._read_cache = [key, batch, None, None, None]   # _train adds the packed batch and the coefficients it found (or its group)
        return batch

    def _read_files(self, input_path, tensor_metadata, schema_params, num_features):
        return read_grouped_partition(
            input_path, tensor_metadata, entity_name=self.model_params.partition_entity,
            feature_bag=self.feature_bag_name, offset_column_name=self.model_params.offset_column_name,
            uid_column_name=schema_params.uid_column_name,
            label_column_name=schema_params.label_column_name, weight_column_name=schema_params.weight_column_name,
            num_features=num_features, wire=self._wants_wire())

    def _wants_wire(self):
        """The reader narrows the partition to the 32-bit hand-over form when a device solver will take it (it uploads that form as it
        is: 0.47 of the bytes of a C2 partition, and no 64-bit arrays for the main thread to pass over); the CPU stand-in of the host
        tests keeps the 64-bit arrays it works on."""
        if os.environ.get("GDMIX_IO_WIRE", "1") == "0":     # A/B switch (tools/r04_wire.sh)
            return False
        s = self._solver      # None: not created yet — it will be the device solver (there is no other in the product path)
        return s is None or isinstance(s, REDeviceSolver)

    def _read_ahead(self, input_path, tensor_metadata, schema_params, num_features):
        """prefetch(): decode the partition and — once the device solver exists — copy its arrays to HBM on a stream of this
        thread's own, so that the main thread finds the partition resident (the copy of a 125 k-entity C2 partition from
        pageable memory is 6 of the ~22 ms the main thread spends per partition: profiles/r04_host_path.txt). The first
        partitions of a run, decoded before the solver exists, are uploaded by the main thread as before."""
        batch = self._read_files(input_path, tensor_metadata, schema_params, num_features)
        s = self._solver
        if isinstance(s, REDeviceSolver) and batch.E > 0 and not self.model_params.rebalance_entities:
            try:
                import torch
                st = self.__dict__.get("_upload_stream")
                if st is None:
                    st = self.__dict__.setdefault("_upload_stream", torch.cuda.Stream(device=s.device))
                with torch.cuda.stream(st):
                    raw = s.upload_wire(batch.to_wire()) if isinstance(batch, WireRawBatch) else s.upload(batch)
                    ev = torch.cuda.Event()
                    ev.record(st)
                batch._device = (raw, ev)
            except Exception as e:    # the main thread uploads it then; a real device problem shows up there
                logger.debug(f"upload ahead of {input_path} failed: {e}")
        return batch

    def _pack(self, solver, batch):
        """gdmix_re_pack of a partition; of its copy in HBM when _read_ahead made one."""
        ahead = batch.__dict__.pop("_device", None) if isinstance(solver, REDeviceSolver) else None
        if ahead is None:
            return solver.pack(batch, has_intercept=self.has_intercept)
        import torch
        raw, ev = ahead
        cur = torch.cuda.current_stream(solver.device)
        cur.wait_event(ev)
        for v in raw.values():       # allocated on the upload stream, used (and later freed) on this one
            if isinstance(v, torch.Tensor):
                v.record_stream(cur)
        return solver.pack(solver.widen(raw) if "ent_n" in raw else raw, has_intercept=self.has_intercept)

    def _train(self, input_path, tensor_metadata, model_weights, num_features, schema_params, output_model_file):
        logger.info(f"Start training with {f'loaded {len(model_weights)} previous models' if model_weights else 'zeros'} "
                    f"as the model initial value.")
        key = self._read_key(input_path, num_features)
        member = None
        if self._group is not None and key in self._group["keys"] and not model_weights:
            member = self._group_member(self._group, key, num_features)      # solved with the partitions around it, or None
        batch = self._read(input_path, tensor_metadata, schema_params, num_features, need_label=True)
        if not batch.has_label:
            raise KeyError(f"label column {schema_params.label_column_name!r} is missing from the training data")
        if member is not None and member["batch"] is batch:
            theta_thr, variance, uniq, feat_ptr, stats = member["solve"]
            resident = None
        else:
            theta_thr, variance, uniq, feat_ptr, stats, resident = self._solve_batch(batch, model_weights, num_features)
        ic = 1 if self.has_intercept else 0
        coef_ptr = feat_ptr + np.arange(batch.E + 1, dtype=np.int64) * ic
        self.last_training_stats = dict(entities=batch.E, samples=batch.N, nnz=batch.Z, **stats)
        results = ModelTable()
        results.add_chunk(batch.entity_ids, theta_thr, coef_ptr, uniq, feat_ptr, variance)
        if self._read_cache is not None and self._read_cache[1] is batch and len(results) == batch.E:   # (an entity id listed twice is scored with its later model: no shortcut)
            if resident is not None:
                self._read_cache[2:4] = list(resident)
            elif member is not None and member["batch"] is batch:
                self._read_cache[4] = (self._group, member["rows"])     # scored from the group's device batch (_predict)
        # The trained model is updated over the prior model: prior entities that are not in the current data
        # are carried over (random_effect_lr_lbfgs_model.py:155-162).
        model_weights.update(results)
        # (the count of the merged table is a hash join of prior and trained ids — 6 ms per 125 k entities: logged by the thread that
        # writes the file, which needs the join anyway, not by this one)
        self._write_behind(output_model_file, self._save_model, output_model_file, model_coefficients=model_weights, num_features=num_features,
                           feature_file=self.feature_file, log_total=True)
        return model_weights

    # ---- several partitions in one device batch (cold start) -----------------------------------------------------
    def plan_group(self, input_paths, metadata_file, schema_params):
        """The driver is about to train these partitions one after another, none of them with a prior model: decode them all ahead;
        the first train() call among them then solves all of them in ONE device batch (their wire forms concatenated in HBM) and
        every call takes its own entities' slice of the result — models, statistics and scores per partition as if each had been solved
        alone (an entity's kernel is chosen per batch, so its last bits may differ from a run partition by partition, as they may
        between any two batch compositions: include/gdmix_re.h). -> the number of partitions taken (0: not grouping)."""
        self._group = None      # (what an unused group decoded stays in _decoded / _prefetched for _read)
        if GROUP_MAX < 2 or len(input_paths) < 2 or self._io_pool is None or self.model_params.rebalance_entities \
                or not self._wants_wire():
            return 0
        if self._solver is None:
            try:
                self._get_solver()      # before the decodes are queued: they upload what they decode once the solver exists
            except Exception:           # no device: the first solve says so, where it always did
                return 0
            if not isinstance(self._solver, REDeviceSolver):
                return 0
        tensor_metadata = DatasetMetadata(read_json_file(metadata_file))
        num_features = 1 if self.feature_bag_name is None else tensor_metadata.get_feature_shape(self.feature_bag_name)[0]
        paths = list(input_paths)[:GROUP_MAX]
        for p in paths:
            self.prefetch(p, metadata_file, schema_params)
        self._group = dict(keys=[self._read_key(p, num_features) for p in paths], paths=paths, state="planned", members={},
                           resident=None, scores=None, meta=(tensor_metadata, schema_params))
        return len(paths)

    @staticmethod
    def group_limits():
        """(partitions, bytes of input files) a group may have: what the driver plans with."""
        return GROUP_MAX, GROUP_MAX_BYTES

    def _group_member(self, g, key, num_features):
        if g["state"] == "planned":
            g["state"] = "failed"       # unless the solve below completes
            self._group_solve(g, num_features)
        return g["members"].get(key) if g["state"] == "solved" else None

    def _device_wire(self, solver, batch):
        """The wire form of a decoded partition in HBM: what _read_ahead uploaded, or uploaded now."""
        ahead = batch.__dict__.pop("_device", None)
        if ahead is not None:
            import torch
            raw, ev = ahead
            cur = torch.cuda.current_stream(solver.device)
            cur.wait_event(ev)
            for v in raw.values():       # allocated on the upload stream, used (and later freed) on this one
                if isinstance(v, torch.Tensor):
                    v.record_stream(cur)
            if "ent_n" in raw:
                return raw
        return solver.upload_wire(batch.to_wire())

    @staticmethod
    def _cat_wire(t, wires):
        """Wire forms of several partitions (device tensors) -> one. Count and index arrays of different widths are widened to the
        widest first. None if the partitions do not go together (weights in some of them only)."""
        out = {k: sum(w[k] for w in wires) for k in ("E", "N", "Z")}
        for k in ("row_nnz_width", "col_width", "y_width"):
            out[k] = max(w[k] for w in wires)
        for k in REDeviceSolver.WIRE_ARRAYS:
            parts = [w[k] for w in wires]
            if all(x is None for x in parts):
                out[k] = None
                continue
            if any(x is None for x in parts):
                return None
            widest = max((x.dtype for x in parts), key=lambda d: t.empty(0, dtype=d).element_size())
            parts = [x if x.dtype == widest else x.to(widest) for x in parts]
            out[k] = t.cat([x.reshape(-1).view(t.uint8) for x in parts]).view(widest)     # (byte views: every width concatenates)
        return out

    def _group_solve(self, g, num_features):
        """Pack + solve of all partitions of the group in one batch; g["members"][key] = that partition's slice."""
        tensor_metadata, schema_params = g["meta"]
        batches = []
        for key, path in zip(g["keys"], g["paths"]):
            b = self._decoded.get(key)
            if b is None:
                ahead = self._prefetched.pop(key, None)
                b = ahead.result() if ahead is not None else self._read_files(path, tensor_metadata, schema_params, num_features)
                self._decoded[key] = b       # _read finds it there when its turn comes
            batches.append(b)
        live = [(k, b) for k, b in zip(g["keys"], batches) if b.E > 0]
        if len(live) < 2 or not all(isinstance(b, WireRawBatch) and b.has_label for _, b in live):
            return
        solver, opts = self._get_solver(), self._solver_options()
        cat = self._cat_wire(solver.torch, [self._device_wire(solver, b) for _, b in live])
        if cat is None:
            return
        pack = lambda: solver.pack(solver.widen(cat), has_intercept=self.has_intercept)
        packed, solved, res = self._pack_and_solve(solver, pack, opts, None)
        E = cat["E"]
        self._check_statuses(res["status"], E)
        ic = 1 if self.has_intercept else 0
        feat_ptr = host_array(packed.ent_feat_ptr())
        uniq = host_array(packed.unique_global().to(torch_int64()))
        coef_ptr = feat_ptr + np.arange(E + 1, dtype=np.int64) * ic
        theta_thr, variance = res["theta_thr"], res.get("variance")
        e0 = n0 = 0
        for key, b in live:
            e1, n1 = e0 + b.E, n0 + b.N
            c0, c1 = int(coef_ptr[e0]), int(coef_ptr[e1])
            g["members"][key] = dict(batch=b, rows=(n0, n1), solve=(
                theta_thr[c0:c1], None if variance is None else variance[c0:c1], uniq[int(feat_ptr[e0]):int(feat_ptr[e1])],
                feat_ptr[e0:e1 + 1] - feat_ptr[e0], {k: res[k][e0:e1] for k in self._STAT_KEYS}))
            e0, n0 = e1, n1
        theta_dev = getattr(solved, "theta_thr", None)
        g["resident"] = (packed, theta_dev if theta_dev is not None else theta_thr)
        g["state"] = "solved"
        logger.info(f"{len(live)} partitions solved in one device batch: {E} entities, {cat['N']} samples")

    def _group_scores(self, g, rows):
        """Scores of the training samples rows[0]:rows[1] of the group's batch with the models just trained: one scoring pass for the
        whole group, at the first partition that asks."""
        if g["scores"] is None:
            packed, theta = g["resident"]
            logit, per_coord = self._get_solver().score(packed, theta, None)
            g["scores"] = (host_array(logit), host_array(per_coord))
            g["resident"] = None        # the packed batch and the coefficients leave HBM
        logit, per_coord = g["scores"]
        return logit[rows[0]:rows[1]], per_coord[rows[0]:rows[1]]

    def _pack_and_solve(self, solver, pack, opts, theta0):
        """pack() -> solve -> results on the host; once more without the tall team class if a team barrier timed out."""
        packed = pack()
        solved = solver.solve(packed, opts, theta0=theta0)
        res = solved.to_host(("theta_thr", "variance") + self._STAT_KEYS)
        if (res["status"] == self.ST_ABORTED).any() and getattr(solver, "tall_team_n", 0) != 0:
            # a team of workgroups gave up waiting for a member (a device too busy, or too small, to keep four whole CUs per
            # team resident: csrc/re_solve_tall.hip). Not a reason to lose the job: the partition again with every tall
            # entity on ONE workgroup (same arithmetic up to the order of its sums), once.
            logger.warning(f"{int((res['status'] == self.ST_ABORTED).sum())} entities timed out at a team barrier: "
                           "solving the partition again without the tall team class")
            keep = solver.tall_team_n
            solver.set_tall_team_n(0)
            try:
                solved = packed = None
                packed = pack()
                solved = solver.solve(packed, opts, theta0=theta0)
                res = solved.to_host(("theta_thr", "variance") + self._STAT_KEYS)
            finally:
                solver.set_tall_team_n(keep)
        return packed, solved, res

    _STAT_KEYS = ("nit", "nfev", "status", "fval", "gnorm")

    def _rebalancing(self, model_weights):
        """-> (entities travel between ranks this round, prior models travel with them). Only when asked for and in a
        multi-rank job. Collective: every rank reaches the same decision; a prior model on any rank makes all ranks
        exchange prior models (a warm start needs the coefficients where the entity is solved)."""
        if not self.model_params.rebalance_entities:
            return False, False
        try:
            import torch.distributed as dist
        except ImportError:
            return False, False
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return False, False
        flags = [None] * dist.get_world_size()
        dist.all_gather_object(flags, bool(model_weights))
        return True, any(flags)

    def _theta0_block(self, count):
        """A page-locked float64 host tensor of at least `count` entries, reused."""
        import torch
        if not torch.cuda.is_available():   # (a solver stand-in in the host tests)
            return None
        blk = getattr(self, "_theta0_stage", None)
        if blk is None or blk.numel() < count:
            blk = torch.empty(int(count * 1.25) + 1024, dtype=torch.float64, pin_memory=True)
            self._theta0_stage = blk
        return blk

    def _solve_batch(self, batch, model_weights, num_features):
        """-> thresholded coefficients, variances|None, global feature index per coefficient, feat_ptr, solver statistics
        for the entities of `batch`, in its order. With re-balancing part of the work is done on other ranks and
        other ranks' entities here (rebalance.py); the arithmetic per entity is the same either way."""
        rebalance, with_prior = self._rebalancing(model_weights)
        if rebalance:
            return self._solve_batch_rebalanced(batch, model_weights, num_features, with_prior)
        solver = self._get_solver()
        opts = self._solver_options()
        packed = theta_dev = None
        ic = 1 if self.has_intercept else 0
        if batch.E == 0:
            theta_thr, variance, uniq = np.zeros(0), None, np.zeros(0, np.int64)
            feat_ptr = np.zeros(1, np.int64)
            stats = {k: np.zeros(0) for k in self._STAT_KEYS}
        else:
            packed = self._pack(solver, batch)
            feat_ptr = host_array(packed.ent_feat_ptr())
            uniq = host_array(packed.unique_global().to(torch_int64()))
            theta0 = self._start_point(model_weights, batch.entity_ids, uniq, feat_ptr, batch.E, num_features)
            first = [packed]
            packed = None
            packed, solved, res = self._pack_and_solve(solver, lambda: first.pop() if first else self._pack(solver, batch), opts, theta0)
            theta_thr, variance = res["theta_thr"], res.get("variance")
            self._check_statuses(res["status"], batch.E)
            theta_dev = getattr(solved, "theta_thr", None)    # still in HBM: what the scoring pass of this partition reads
            stats = {k: res[k] for k in self._STAT_KEYS}
        resident = None if packed is None else (packed, theta_dev if theta_dev is not None else theta_thr)
        return theta_thr, variance, uniq, feat_ptr, stats, resident

    def _start_point(self, model_weights, entity_ids, uniq, feat_ptr, E, num_features):
        """theta0 of a warm start in the packed batch's local order (None without prior models)."""
        if not model_weights:
            return None
        ic = 1 if self.has_intercept else 0
        # the starting point goes up from a page-locked block kept from partition to partition (a fresh 64 MB array per
        # partition is 16 k page faults and a staged copy: 20 ms per 125 k entities)
        stage = self._theta0_block(int(feat_ptr[-1]) + E * ic)
        theta0, _ = _model_coefficients_for_batch(model_weights, entity_ids, uniq, feat_ptr, self.has_intercept, num_features,
                                                  out=None if stage is None else stage.numpy())
        if stage is not None:
            theta0 = stage[:int(feat_ptr[-1]) + E * ic]
        return theta0

    ST_ABORTED = 9      # GDMIX_RE_ST_ABORTED (include/gdmix_re.h)

    @staticmethod
    def _check_statuses(status, E):
        bad = (status < 0) | (status > 4)      # 0..4 are fmin_l_bfgs_b's own outcomes
        if bad.any():
            raise RuntimeError(f"{int(bad.sum())} of {E} entities were not solved (device status "
                               f"{sorted(set(np.asarray(status)[bad].tolist()))}: 9 = a team barrier timed out, -1 = never reached)")

    def _solve_batch_rebalanced(self, batch, model_weights, num_features, with_prior):
        """_solve_batch with entity re-balancing (rebalance.py): the partition's 32-bit wire form goes to the device, the
        travelling entities are exchanged device to device, kept + received entities are widened, packed and solved in HBM,
        the results travel back device to device and are read back once, in this partition's entity order. With the CPU
        stand-in of the host tests the same exchange runs on CPU tensors."""
        import torch
        from .rebalance import Rebalancer, wire_tensors, wire_to_raw
        solver = self._get_solver()
        opts = self._solver_options()
        ic = 1 if self.has_intercept else 0
        on_device = hasattr(solver, "widen")
        dev = solver.device if on_device else torch.device("cpu")
        # what the rounds so far measured prices this round's entities (rebalance.SizeCostModel: milliseconds per non-zero by entity
        # size, summed over the workers); the first round, and the CPU stand-in of the host tests, price by non-zeros
        scm = self.__dict__.get("_size_cost")
        nnz = batch.ent_nnz()
        if on_device:
            solver.set_timing(True)
        rb = Rebalancer(batch.ent_n(), nnz, wire_tensors(batch, dev, solver if on_device else None),
                        cost=None if scm is None else scm.cost(nnz), order=None if scm is None else scm.order(nnz))
        prior = model_weights.rows_for(batch.entity_ids) if (with_prior and model_weights) else None
        work = rb.exchange(prior=prior, with_prior=with_prior)
        E_work = work["E"]
        logger.info(f"re-balancing: loads {rb.loads.tolist()}, sent {[int(x.size) for x in rb.sent]}, "
                    f"received {rb.recv_counts}, solving {E_work} entities here")
        as_t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
        if E_work == 0:
            cc = torch.zeros(0, dtype=torch.int64, device=dev)
            th = torch.zeros(0, dtype=torch.float64, device=dev)
            va = torch.zeros(0, dtype=torch.float64, device=dev) if self.model_params.random_effect_variance_mode is not None else None
            fi = torch.zeros(0, dtype=torch.int64, device=dev)
            ints = torch.zeros((0, 3), dtype=torch.int32, device=dev)
            flts = torch.zeros((0, 2), dtype=torch.float64, device=dev)
        else:
            if on_device:
                packed = solver.pack(solver.widen(work), has_intercept=self.has_intercept)
                fp_dev, uq_dev = packed.ent_feat_ptr(), packed.unique_global().to(torch.int64)      # (the exchange's index columns are int64 on every rank)
            else:
                packed = solver.pack(wire_to_raw(work), has_intercept=self.has_intercept)
                fp_dev, uq_dev = as_t(packed.ent_feat_ptr().cpu().numpy(), torch.int64), as_t(packed.unique_global().cpu().numpy(), torch.int64)
            theta0 = None
            if with_prior:   # the models of the entities solved here, wherever they came from, by position
                ids = [str(i) for i in range(E_work)]
                theta0 = self._start_point(ModelTable.from_rows(ids, rb.work_prior), ids, host_array(uq_dev), host_array(fp_dev), E_work,
                                           num_features)
            solved = solver.solve(packed, opts, theta0=theta0)
            if on_device:
                th, va = solved.theta_thr, getattr(solved, "variance", None)
                ints = torch.stack([solved.nit, solved.nfev, solved.status], dim=1)
                flts = torch.stack([solved.fval, solved.gnorm], dim=1)
            else:
                r = solved.to_host(("theta_thr", "variance") + self._STAT_KEYS)
                th = as_t(r["theta_thr"], torch.float64)
                va = None if r.get("variance") is None else as_t(r["variance"], torch.float64)
                ints = torch.stack([as_t(r[k], torch.int32) for k in ("nit", "nfev", "status")], dim=1)
                flts = torch.stack([as_t(r[k], torch.float64) for k in ("fval", "gnorm")], dim=1)
            cc = (fp_dev[1:] - fp_dev[:-1]) + ic
            fi = uq_dev
        if on_device:
            # this round's class times, attributed to the entities solved HERE by size, summed over the workers (every worker joins,
            # also one that solved nothing)
            import torch.distributed as dist
            from .rebalance import SizeCostModel
            from .solver import NUM_CLASSES
            tot = np.zeros((2, SizeCostModel.BUCKETS))
            if E_work > 0:
                znz = packed.ent_nnz_ptr()
                tot = SizeCostModel.totals(host_array(packed._view(packed.c.cls_tmp, packed.E, torch.int32)), host_array(znz[1:] - znz[:-1]),
                                           np.array(solver.last_solve_ms()), NUM_CLASSES)
            tt = torch.from_numpy(tot)
            if dist.get_backend() == "nccl":
                td = tt.to(dev)
                dist.all_reduce(td)
                tt = td.cpu()
            else:
                dist.all_reduce(tt)
            self._size_totals = self.__dict__.get("_size_totals", 0.0) + tt.numpy()
            self._size_cost = SizeCostModel.from_totals(self._size_totals)
        self.last_exchange_devices = (str(work["val"].device), str(th.device))   # (tests: the payload never left the device)
        my_cc, theta_thr, variance, uniq, ints, flts = rb.give_back(cc, th, va, fi, ints, flts, has_intercept=self.has_intercept)
        my_cc, theta_thr, uniq = host_array(my_cc), host_array(theta_thr), host_array(uniq)
        variance = None if variance is None else host_array(variance)
        ints, flts = host_array(ints), host_array(flts)
        self._check_statuses(ints[:, 2], batch.E)
        feat_ptr = np.concatenate([[0], np.cumsum(my_cc - ic)]).astype(np.int64)
        stats = dict(nit=ints[:, 0].astype(np.int32), nfev=ints[:, 1].astype(np.int32), status=ints[:, 2].astype(np.int32),
                     fval=flts[:, 0].copy(), gnorm=flts[:, 1].copy())
        return theta_thr, variance, uniq.astype(np.int64), feat_ptr, stats, None

    def idle_round(self, num_features=1):
        """A rank without a partition in this round still takes part in the re-balancing collectives (and solves what
        it is sent)."""
        if not self.model_params.rebalance_entities:
            return
        from .batch import RawBatch
        z = lambda dt: np.zeros(0, dt)
        empty = RawBatch(ent_row_ptr=np.zeros(1, np.int64), row_nnz_ptr=np.zeros(1, np.int64), col_global=z(np.int64),
                         val=z(np.float32), y=z(np.float32), offset=z(np.float32), weight=None, uid=z(np.int64),
                         entity_ids=[], has_label=True)
        self._solve_batch(empty, ModelTable(), num_features)

    def _predict(self, input_path, tensor_metadata, output_file, schema_params, num_features, model_weights):
        logger.info(f"Start inference for {input_path}.")
        packed = theta = has_model = group = None
        cache = self._read_cache
        if cache is not None and cache[0] == self._read_key(input_path, num_features) and (cache[2] is not None or cache[4] is not None):
            # The partition that was just trained on: still packed on the device (alone, or inside the batch of its group), and every
            # entity's model is the one the solve returned (thresholded, as saved) - what the table lookup below would reproduce
            # coefficient by coefficient.
            _, batch, packed, theta, group = cache
            self._read_cache = None
        else:
            batch = self._read(input_path, tensor_metadata, schema_params, num_features, need_label=False)
        has_weight = any(schema_params.weight_column_name == f.name for f in tensor_metadata.get_features())
        schema = avro.inference_output_schema(schema_params, has_weight=has_weight)
        if batch.E == 0:
            avro.write_file(output_file, schema, [])
            return
        solver = self._get_solver()
        if group is not None:
            logit, per_coord = self._group_scores(*group)
        else:
            if packed is None:
                packed = self._pack(solver, batch)
                feat_ptr = host_array(packed.ent_feat_ptr())
                uniq = host_array(packed.unique_global().to(torch_int64()))
                theta, has_model = _model_coefficients_for_batch(model_weights, batch.entity_ids, uniq, feat_ptr,
                                                                 self.has_intercept, num_features)
            logit, per_coord = solver.score(packed, theta, has_model)
            logit, per_coord = host_array(logit), host_array(per_coord)
        weights = batch.weight if batch.weight is not None else np.ones(batch.N, np.float32)
        self._write_behind(output_file, _write_scores, output_file, schema, schema_params, batch.uid, logit, batch.y if batch.has_label else None,
                           weights if has_weight else None, per_coord)
        logger.info(f"Inference complete: {input_path}.")

    def _encoded_features(self, feature_file):
        """(feature list, its Avro-encoded form) of a feature file, read and encoded once per model object."""
        cache = self.__dict__.setdefault("_feature_cache", {})
        if feature_file not in cache:
            fast = native_reader.EncodedFeatures.from_feature_file(feature_file) if native_reader.available() else None
            if fast is not None:       # (the list of (name, term) tuples is read only if somebody iterates over it: nobody does on the native path)
                cache[feature_file] = (_LazyFeatureList(feature_file), fast)
            else:
                fl = read_feature_list(feature_file)
                enc = [avro.enc_string(n) + avro.enc_string(t) for (n, t) in fl]
                cache[feature_file] = (fl, native_reader.EncodedFeatures(enc) if native_reader.available() else enc)
        return cache[feature_file]

    def _save_model(self, output_file, model_coefficients, num_features, feature_file, log_total=False):
        if log_total:
            logger.info(f"{len(model_coefficients)} models in total after training/refreshing.")
        feature_list = self._encoded_features(feature_file) if feature_file else None
        if feature_file is None:
            assert num_features == 1   # intercept only model
        with_variance = self.model_params.random_effect_variance_mode is not None
        n = _export_models_to_avro(output_file, model_coefficients, feature_list, self.has_intercept, with_variance,
                                   sparsity_threshold=1.0e-4)   # export threshold is always the default
        logger.info(f"dumped {n} models to avro file at {output_file}.")

    def _load_weights(self, model_file, catch_exception=False):
        logger.info(f"Loading model from {model_file}")
        ahead = self._prefetched_models.pop(os.path.abspath(model_file), None)
        if ahead is not None:
            return ahead.result()
        self.flush(model_file)    # it may be a file this process is still writing
        if not os.path.exists(model_file):
            if catch_exception:
                logger.info(f"No model found at {model_file}.")
                return ModelTable()
            raise FileNotFoundError(f"Model file {model_file} does not exist")
        return self._load_weights_from(model_file)

    def _load_weights_from(self, model_file):
        table = ModelTable()
        if self.feature_file is not None and native_reader.available():
            loaded = self._load_weights_native(model_file, table)
            if loaded:
                return table
        feature2global_id = None if self.feature_file is None else get_feature_map(self.feature_file)
        ids, theta, var, idx, coef_ptr, feat_ptr = [], [], [], [], [0], [0]
        any_var = False
        for record in avro.read_file(model_file):
            mid, tr = self._convert_avro_model_record_to_sparse_coefficients(self.has_intercept, record, feature2global_id)
            ids.append(mid)
            theta.append(tr.theta)
            idx.append(tr.unique_global_indices)
            if tr.variance is not None:
                any_var = True
                var.append(tr.variance)
            else:
                var.append(np.zeros(len(tr.theta)))
            coef_ptr.append(coef_ptr[-1] + len(tr.theta))
            feat_ptr.append(feat_ptr[-1] + len(tr.unique_global_indices))
        if ids:
            table.add_chunk(ids, np.concatenate(theta), coef_ptr, np.concatenate(idx) if idx else np.zeros(0, np.int64),
                            feat_ptr, np.concatenate(var) if any_var else None)
        return table

    def _load_weights_native(self, model_file, table):
        """The same table through libgdmix_io.so (blocks decoded in parallel); False when the file's writer schema is not
        the canonical layout, in which case the schema-driven Python decoder reads it."""
        schema, codec, sync, data_offset = avro.read_header(model_file)
        if codec not in ("null", "deflate") or not avro.is_model_schema(schema):
            return False
        prefix = self._encoded_features(self.feature_file)[1]
        icpt = avro.enc_string(constants.INTERCEPT) + avro.enc_string("")
        m = native_reader.read_models_avro(model_file, data_offset, sync, codec == "deflate", prefix, icpt, self.has_intercept)
        if m["ids"]:
            ic = 1 if self.has_intercept else 0
            coef_ptr = m["coef_ptr"]
            feat_ptr = coef_ptr - np.arange(coef_ptr.size, dtype=np.int64) * ic
            table.add_chunk(m["ids"], m["mean"], coef_ptr, m["feat_idx"], feat_ptr, m["variance"])
        return True

    @staticmethod
    def _convert_avro_model_record_to_sparse_coefficients(has_intercept, model_record, feature2global_id):
        """random_effect_lr_lbfgs_model.py:275-309."""
        model_id = model_record["modelId"]
        coeffs, uidx, variance = [], [], []
        for i, ntv in enumerate(model_record["means"]):
            coeffs.append(np.float64(ntv["value"]))
            if has_intercept and i == 0:
                assert ntv["name"] == constants.INTERCEPT and ntv["term"] == ""
            else:
                uidx.append(feature2global_id[(ntv["name"], ntv["term"])])
        if model_record.get("variances"):
            for i, ntv in enumerate(model_record["variances"]):
                variance.append(np.float64(ntv["value"]))
                if has_intercept and i == 0:
                    assert ntv["name"] == constants.INTERCEPT and ntv["term"] == ""
                else:
                    assert uidx[i - 1] == feature2global_id[(ntv["name"], ntv["term"])]
        if feature2global_id is None:
            # intercept-only model, add one dummy feature
            assert len(uidx) == 0
            coeffs.append(np.float64(0.0))
            uidx.append(0)
            if variance:
                variance.append(np.float64(0.0))
        return model_id, TrainingResult(theta=np.array(coeffs), variance=np.array(variance) if variance else None,
                                        unique_global_indices=np.array(uidx, np.int64))


class _LazyFeatureList:
    """The (name, term) list of a feature file, read when first asked for."""

    def __init__(self, path):
        self._path, self._list = path, None

    def _get(self):
        if self._list is None:
            self._list = read_feature_list(self._path)
        return self._list

    def __iter__(self):
        return iter(self._get())

    def __len__(self):
        return len(self._get())

    def __getitem__(self, i):
        return self._get()[i]


# ---- Avro writers (record layout of util/io_utils.py:102-212 and :299-375) ----------------------------------
def _export_models_to_avro(output_file, table, feature_list, has_intercept, with_variance, sparsity_threshold=1e-4,
                           native=None, sync_marker=None, model_class=None):
    """One BayesianLinearModelAvro per entity: intercept always, features with |value| > threshold
    (gen_one_avro_model, util/io_utils.py:102-160). The array payloads are assembled from pre-encoded
    name/term prefixes instead of per-field schema dispatch."""
    model_class = model_class or constants.PHOTON_LR_MODEL_CLASS
    head_class = avro.enc_long(1) + avro.enc_string(model_class)       # union branch 1 (string)
    loss = avro.enc_long(1) + avro.enc_string("")                      # lossFunction = "" (not null)
    icpt = avro.enc_string(constants.INTERCEPT) + avro.enc_string("")
    prefix = None
    if isinstance(feature_list, tuple) and len(feature_list) == 2 and not isinstance(feature_list[0], tuple):
        feature_list, prefix = feature_list        # (list, encoded form) from RandomEffectLRLBFGSModel._encoded_features
    elif feature_list is not None:
        prefix = [avro.enc_string(n) + avro.enc_string(t) for (n, t) in feature_list]
    if native is None:
        native = native_reader.available() and isinstance(table, ModelTable)
    if native:
        ids, coef_beg, coef_cnt, var_beg, feat_beg, mean, variance, idx = table.flatten()
        header, sync = avro.container_header(avro.BAYESIAN_LINEAR_MODEL_SCHEMA, "null", sync_marker)
        use_var = with_variance and variance is not None
        return native_reader.write_models_avro(output_file, header, sync, ids, coef_beg, coef_cnt, mean, feat_beg, idx, prefix,
                                               icpt, head_class, loss, has_intercept, sparsity_threshold,
                                               var_beg if use_var else None, variance if use_var else None)
    pack_d = struct.Struct("<d").pack
    ic = 1 if has_intercept else 0
    count = 0
    with avro.Writer(output_file, avro.BAYESIAN_LINEAR_MODEL_SCHEMA, sync_marker=sync_marker) as w:
        buf = bytearray()
        nbuf = 0
        for model_id, (mean, variance, uidx) in table.items():
            has_var = with_variance and variance is not None
            items, vitems = [], []
            if ic:
                items.append(icpt + pack_d(float(mean[0])))
                if has_var:
                    vitems.append(icpt + pack_d(float(variance[0])))
            if prefix is not None:
                vals = mean[ic:]
                keep = np.flatnonzero(np.abs(vals) > sparsity_threshold)
                for k in keep:
                    items.append(prefix[int(uidx[k])] + pack_d(float(vals[k])))
                    if has_var:
                        vitems.append(prefix[int(uidx[k])] + pack_d(float(variance[ic + k])))
            buf += avro.enc_string(str(model_id)) + head_class
            buf += (avro.enc_long(len(items)) + b"".join(items) if items else b"") + avro.enc_long(0)
            if has_var:
                buf += avro.enc_long(1) + (avro.enc_long(len(vitems)) + b"".join(vitems) if vitems else b"") + avro.enc_long(0)
            else:
                buf += avro.enc_long(0)
            buf += loss
            nbuf += 1
            count += 1
            if nbuf >= 1024:
                w.write_encoded(bytes(buf), nbuf)
                buf, nbuf = bytearray(), 0
        if nbuf:
            w.write_encoded(bytes(buf), nbuf)
    return count


def _write_scores(output_file, schema, schema_params, uid, score, label, weight, per_coord, native=None, sync_marker=None):
    """Records {uid long, predictionScore float, label [null,float], weight float?, perCoordinate float} in
    blocks of 1024 (batched_write_avro, util/io_utils.py:299-334)."""
    if native is None:
        native = native_reader.available()
    if native:
        header, sync = avro.container_header(schema, "null", sync_marker)
        native_reader.write_scores_avro(output_file, header, sync, uid, score, label, weight, per_coord)
        return
    pack_f = struct.Struct("<f").pack
    n = len(uid)
    with avro.Writer(output_file, schema, sync_marker=sync_marker) as w:
        for b0 in range(0, n, 1024):
            b1 = min(n, b0 + 1024)
            buf = bytearray()
            for i in range(b0, b1):
                buf += avro.enc_long(int(uid[i])) + pack_f(float(score[i]))
                if label is None:
                    buf += b"\x00"
                else:
                    buf += b"\x02" + pack_f(float(label[i]))
                if weight is not None:
                    buf += pack_f(float(weight[i]))
                buf += pack_f(float(per_coord[i]))
            w.write_encoded(bytes(buf), b1 - b0)
