"""FixedEffectLRModelLBFGS on an MI355X: the Python mirror of the reference's fixed-effect model class
(gdmix-trainer/src/gdmix/models/custom/fixed_effect_lr_lbfgs_model.py:57-812) around the device solver of
gdmix_amd/fixed_effect.py. Same constructor, attributes, train / predict / export signatures and files:

  in   <training_data_dir>/*.tfrecord[.gz|.deflate]   one tf.train.Example per sample (per_record_input_fn,
       io/input_data_pipeline.py:129-221): dense columns (uid, label, offset, weight) as scalars, the sparse bag as
       `<bag>_indices` / `<bag>_values`; files are sorted and worker w of W reads files[w::W] (or file w when there
       are fewer files than workers; util/distribution_utils.py:11-47)
  out  <output_model_dir>/part-00000.avro             one BayesianLinearModelAvro, modelId "global model" (:690-728),
       written by the chief after thresholding |theta| <= 1e-4 -> 0 (:648-649)
       <training_score_dir>/part-{task:05d}.avro, <validation_score_dir>/part-{task:05d}.avro   (:406-440)

Training with W > 1 workers: every worker runs this with torch.distributed initialised; gradient and value are
all-reduced once per L-BFGS evaluation (fixed_effect.py: run_stepping_loop), the step is replicated.
Not carried over: fixed_effect_variance_mode (rejected), copy_to_local and the TF server knobs (accepted, unused).
"""
import glob
import logging
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import constants
from .fixed_effect import LINEAR_REGRESSION, LOGISTIC_REGRESSION, FixedEffectDeviceSolver
from .io import avro, native_reader, tfrecord
from .io.features import read_feature_list
from .io.grouped_reader import resolve_input_files
from .io.metadata import DatasetMetadata
from .params import LRParams

logger = logging.getLogger(__name__)
logger.setLevel(logging.INFO)

GLOBAL_MODEL_ID = "global model"
MODEL_CLASS = {LOGISTIC_REGRESSION: "com.linkedin.photon.ml.supervised.classification.LogisticRegressionModel",
               LINEAR_REGRESSION: "com.linkedin.photon.ml.supervised.regression.LinearRegressionModel"}


@dataclass
class FixedLRParams(LRParams):
    """fixed_effect_lr_lbfgs_model.py:57-71."""
    copy_to_local: bool = True
    num_server_creation_retries: int = 50
    retry_interval: int = 2
    delayed_exit_in_seconds: int = 60
    disable_fixed_effect_scoring_after_training: bool = False
    fixed_effect_variance_mode: Optional[str] = None

    def __post_init__(self):
        super().__post_init__()
        assert self.fixed_effect_variance_mode is None or self.fixed_effect_variance_mode in (constants.FULL, constants.SIMPLE), \
            f"Action: {self.fixed_effect_variance_mode} must be in {(constants.FULL, constants.SIMPLE)}"


def shard_input_files(input_path, num_shards, shard_index):
    """util/distribution_utils.py:11-47: every entry of the directory (glob '*': whatever its suffix, dot-files
    excluded), sorted, strided over the workers; with fewer files than workers, worker w gets file w (or nothing).
    Entries that are not record files (Spark's zero-byte `_SUCCESS` marker, sub-directories: upstream's glob returns them too)
    take part in the striding exactly as they do upstream and then contribute no records (read_per_record_files)."""
    assert 0 <= shard_index < num_shards and num_shards >= 1
    pattern = os.path.join(input_path, "*") if os.path.isdir(input_path) else input_path
    files = sorted(glob.glob(pattern))
    assert len(files) > 0, f"{input_path} is empty"
    if len(files) < num_shards:
        return [files[shard_index]] if shard_index < len(files) else []
    return files[shard_index::num_shards]


def read_per_record_files(files, metadata: DatasetMetadata, feature_bag, num_features, uid_name, label_name, offset_name,
                          weight_name, native=None):
    """tf.train.Example records -> flat sample arrays (CSR over samples). Columns that the metadata does not list are
    defaults: offset 0, weight 1, label 0 (fixed_effect_lr_lbfgs_model.py:255-258,345-346). native None: libgdmix_io.so
    when built (same rules; tests/test_fe_model.py compares the two)."""
    files = [f for f in files if os.path.isfile(f) and os.path.getsize(f) > 0]   # zero-byte entries (`_SUCCESS`) and directories hold no records
    names = set(metadata.get_feature_names()) | set(metadata.get_label_names())
    has = lambda n: n is not None and n in names
    has_label, has_offset, has_weight = has(label_name), has(offset_name), has(weight_name)
    if native is None:
        native = native_reader.available()
    if native:
        d = native_reader.read_example_files(files, feature_bag, num_features, uid_name, label_name if has_label else None,
                                             offset_name if has_offset else None, weight_name if has_weight else None)
        d["has_label"], d["has_weight"] = has_label, has_weight
        return d
    uid, y, off, w, k, cols, vals = [], [], [], [], [], [], []
    ikey, vkey = (f"{feature_bag}_indices", f"{feature_bag}_values") if feature_bag else (None, None)

    def scalar(feats, name, what):
        if name not in feats:
            raise KeyError(f"column {name!r} is missing from a record")
        kind, v = feats[name]
        if len(v) != 1:
            raise ValueError(f"{what} column {name!r} must hold one value per record, got {len(v)}")
        return v[0]

    for fn in files:
        for rec in tfrecord.iter_records(fn):
            feats = tfrecord.decode_example(rec)
            uid.append(int(scalar(feats, uid_name, "uid")))
            y.append(float(scalar(feats, label_name, "label")) if has_label else 0.0)
            off.append(float(scalar(feats, offset_name, "offset")) if has_offset else 0.0)
            w.append(float(scalar(feats, weight_name, "weight")) if has_weight else 1.0)
            if feature_bag:
                ci = feats.get(ikey, ("int64", np.zeros(0, np.int64)))[1]
                cv = feats.get(vkey, ("float", np.zeros(0, np.float32)))[1]
                if len(ci) != len(cv):
                    raise ValueError(f"{ikey} and {vkey} differ in length in a record of {fn}")
                ci = np.asarray(ci, np.int64)
                if ci.size and (ci.min() < 0 or ci.max() >= num_features):
                    raise ValueError(f"feature index outside [0, {num_features}) in {fn}")
                k.append(len(ci))
                cols.append(ci)
                vals.append(np.asarray(cv, np.float32))
            else:
                k.append(0)
    n = len(uid)
    cat = lambda parts, dt: np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)
    return dict(n=n, row_nnz_ptr=np.concatenate([[0], np.cumsum(np.array(k, np.int64))]).astype(np.int64),
                col=cat(cols, np.int64), val=cat(vals, np.float32), y=np.array(y, np.float32), offset=np.array(off, np.float32),
                weight=np.array(w, np.float32), uid=np.array(uid, np.int64), has_label=has_label, has_weight=has_weight)


class FixedEffectLRModelLBFGS:
    """Linear / logistic regression over the whole data set, trained on the device."""

    def __init__(self, raw_model_params, base_training_params, device=None):
        self.model_params: FixedLRParams = self._parse_parameters(raw_model_params)
        p = self.model_params
        self.fixed_effect_variance_mode = p.fixed_effect_variance_mode
        self.variances = None
        self.training_output_dir = base_training_params.training_score_dir
        self.validation_output_dir = base_training_params.validation_score_dir
        self.model_type = base_training_params.model_type
        self.training_data_dir = p.training_data_dir
        self.validation_data_dir = p.validation_data_dir
        self.metadata_file = p.metadata_file
        self.checkpoint_path = p.output_model_dir
        self.data_format = p.data_format
        self.offset_column_name = p.offset_column_name
        self.feature_bag_name = p.feature_bag
        self.feature_file = p.feature_file if self.feature_bag_name else None
        self.has_intercept = p.has_intercept
        self.is_regularize_bias = p.regularize_bias
        self.max_iteration = p.num_of_lbfgs_iterations
        self.l2_reg_weight = p.l2_reg_weight
        self.sparsity_threshold = p.sparsity_threshold
        self.num_correction_pairs = p.num_of_lbfgs_curvature_pairs
        if self.model_type == constants.LOGISTIC_REGRESSION:
            self.disable_fixed_effect_scoring_after_training = p.disable_fixed_effect_scoring_after_training
        else:   # no inference after training for plain linear regression (:110-113)
            self.disable_fixed_effect_scoring_after_training = True
        self.metadata = DatasetMetadata(self.metadata_file)
        self.num_features = self._get_num_features()
        self.model_coefficients = None
        self.last_training_info = None
        self._device = device
        self._fe = None

    # ---- helpers ------------------------------------------------------------------------------------------------
    def _parse_parameters(self, raw_model_parameters):
        return FixedLRParams.__from_argv__(raw_model_parameters, error_on_unknown=False)

    def _get_num_features(self):
        if self.feature_bag_name is None:
            return 1   # intercept only model: a dummy feature (:157-165)
        return self.metadata.get_feature_shape(self.feature_bag_name)[0]

    def _solver(self):
        if self._fe is None:
            device = self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0"))
            self._fe = FixedEffectDeviceSolver(device)
        return self._fe

    def _read(self, input_path, num_workers, task_index, schema_params):
        files = shard_input_files(input_path, num_workers, task_index)
        return read_per_record_files(files, self.metadata, self.feature_bag_name, self.num_features,
                                     schema_params.uid_column_name, schema_params.label_column_name, self.offset_column_name,
                                     schema_params.weight_column_name)

    # ---- train ----------------------------------------------------------------------------------------------------
    def train(self, training_data_dir, validation_data_dir, metadata_file, checkpoint_path, execution_context, schema_params):
        task_index = execution_context[constants.TASK_INDEX]
        num_workers = execution_context[constants.NUM_WORKERS]
        is_chief = execution_context[constants.IS_CHIEF]
        data = self._read(training_data_dir, num_workers, task_index, schema_params)
        prev_model = self._load_model(catch_exception=True)
        expected = self.num_features + 1 if self.has_intercept else self.num_features
        x0 = None
        if prev_model is not None and len(prev_model) == expected:
            logger.info("Found a previous model, loaded as the initial point for training")
            x0 = np.asarray(prev_model, np.float64)
        elif prev_model is not None:
            logger.info(f"Initial model size is {len(prev_model)}, expected {expected}, use all zeros instead.")
        D = self.num_features
        bag = self.feature_bag_name is not None
        theta, info = self._solver().fit_stepping(
            data["row_nnz_ptr"] if bag else np.zeros(data["n"] + 1, np.int64), data["col"] if bag else [], data["val"] if bag else [],
            data["y"], D, offset=data["offset"], weight=data["weight"] if data["has_weight"] else None,
            has_intercept=self.has_intercept, l2=self.l2_reg_weight, regularize_bias=self.is_regularize_bias,
            model_type=self.model_type, theta0=self._strip_dummy(x0) if not bag else x0, max_iter=self.max_iteration,
            m=self.num_correction_pairs, tolerance=self.model_params.lbfgs_tolerance, dummy=not bag,
            variance_mode=self.fixed_effect_variance_mode, threshold=self.sparsity_threshold)
        self.variances = info.pop("variances", None)
        if not bag:
            theta = np.concatenate([[0.0], theta])   # the dummy weight of an intercept-only model (add_dummy_weight)
            if self.variances is not None:
                self.variances = np.concatenate([[0.0], self.variances])
        self.last_training_info = info
        logger.info(f"f_min: {info['fval']} num of funcalls: {info['nfev']} status: {info['status']}")
        theta = np.where(np.abs(theta) <= self.sparsity_threshold, 0.0, theta)   # threshold_coefficients (:648-649)
        self.model_coefficients = theta
        # the reference's variance computation rides on the scoring pass over the training data, which therefore runs (and writes
        # its scores) whenever a variance mode is set (:650-661)
        if not self.disable_fixed_effect_scoring_after_training or self.fixed_effect_variance_mode is not None:
            self._score_and_write(theta, data, task_index, schema_params, self.training_output_dir)
        if validation_data_dir:
            vdata = self._read(validation_data_dir, num_workers, task_index, schema_params)
            self._score_and_write(theta, vdata, task_index, schema_params, self.validation_output_dir)
        if is_chief:
            self._save_model()

    @staticmethod
    def _strip_dummy(x0):
        return None if x0 is None else x0[1:]

    # ---- scoring ---------------------------------------------------------------------------------------------------
    def _score_and_write(self, theta, data, task_index, schema_params, output_dir):
        """logits = X w + b (per-coordinate score), + offset (score); :214-306,406-440."""
        from .fixed_effect import shard_as_batch, to_local
        n = data["n"]
        bag = self.feature_bag_name is not None
        if n == 0:
            per_coord = np.zeros(0, np.float32)
        else:
            fe = self._solver()
            if hasattr(fe, "score"):      # the device path: one pass over the sample-major arrays, no pack
                _, per_coord = fe.score(data["row_nnz_ptr"] if bag else None, data["col"] if bag else None, data["val"] if bag else None,
                                        data["offset"], theta if bag else theta[1:], self.num_features if bag else 0, self.has_intercept)
            else:
                batch, dummy = shard_as_batch(data["row_nnz_ptr"] if bag else np.zeros(n + 1, np.int64), data["col"] if bag else [],
                                              data["val"] if bag else [], np.zeros(n, np.float32), data["offset"], None, self.has_intercept,
                                              dummy=not bag)
                packed = fe.solver.pack(batch, has_intercept=self.has_intercept)
                uniq = packed.unique_global().cpu().numpy()
                th = theta if bag else theta[1:]
                local = to_local(th, uniq, self.num_features if bag else 0, self.has_intercept, dummy)
                logit, per = fe.solver.score(packed, local)
                per_coord = per.cpu().numpy()[:n]    # (a shard without any non-zero carries one padding sample of weight 0)
        score = (per_coord.astype(np.float64) + data["offset"].astype(np.float64)).astype(np.float32)
        self._write_inference_result(data["uid"], data["y"] if data["has_label"] else None,
                                     data["weight"] if data["has_weight"] else None, score, per_coord, task_index, schema_params,
                                     output_dir)

    def _write_inference_result(self, sample_ids, labels, weights, prediction_score, prediction_score_per_coordinate, task_index,
                                schema_params, output_dir):
        schema = avro.inference_output_schema(schema_params, has_weight=weights is not None)
        output_file = os.path.join(output_dir, f"part-{task_index:05d}.avro")
        # the reference writes int(weight) into the float field (:427-428)
        w = None if weights is None else np.trunc(weights).astype(np.float32)
        header, sync = avro.container_header(schema, "null")
        if native_reader.available():
            native_reader.write_scores_avro(output_file, header, sync, sample_ids, prediction_score, labels, w,
                                            prediction_score_per_coordinate)
        else:
            from .model import _write_scores
            _write_scores(output_file, schema, schema_params, sample_ids, prediction_score, labels, w,
                          prediction_score_per_coordinate, native=False)
        logger.info(f"Worker {task_index} has written inference result to {output_file}")

    # ---- model file -------------------------------------------------------------------------------------------------
    def _save_model(self):
        """One BayesianLinearModelAvro: (INTERCEPT) first, then the features with |value| > threshold (:690-728,
        util/io_utils.py:102-160)."""
        from .model import ModelTable, _export_models_to_avro
        theta = self.model_coefficients
        D = self.num_features
        ic = 1 if self.has_intercept else 0
        bag = self.feature_bag_name is not None
        weights = theta[:D] if bag else np.zeros(0)
        local = np.concatenate([theta[D:D + ic], weights])   # intercept first, as the export helper expects
        var_local = None
        if self.variances is not None:
            var_local = np.concatenate([self.variances[D:D + ic], self.variances[:D] if bag else np.zeros(0)])
        table = ModelTable()
        table.add_chunk([GLOBAL_MODEL_ID], local, [0, local.size], np.arange(weights.size, dtype=np.int64), [0, weights.size],
                        variance=var_local)
        feature_list = self._feature_prefixes() if self.feature_file else None
        output_file = os.path.join(self.checkpoint_path, "part-00000.avro")
        _export_models_to_avro(output_file, table, feature_list, self.has_intercept, self.variances is not None, self.sparsity_threshold,
                               model_class=MODEL_CLASS[self.model_type])
        logger.info(f"dumped the global model to {output_file}")

    def _feature_prefixes(self):
        """(feature list, its Avro-encoded form) as _export_models_to_avro and the native model reader take them; for a plain feature
        file straight from its bytes (native_reader.EncodedFeatures.from_feature_file: 100 k features in milliseconds instead of 0.2 s)."""
        if native_reader.available():
            fast = native_reader.EncodedFeatures.from_feature_file(self.feature_file)
            if fast is not None:
                return (None, fast)
        fl = read_feature_list(self.feature_file)
        enc = [avro.enc_string(n) + avro.enc_string(t) for (n, t) in fl]
        return (fl, native_reader.EncodedFeatures(enc) if native_reader.available() else enc)

    def _load_model(self, catch_exception=False):
        """-> coefficients [num_features (+1, intercept last)] or None (:730-747, load_linear_models_from_avro)."""
        if not (self.checkpoint_path and os.path.exists(self.checkpoint_path)):
            if catch_exception:
                return None
            raise FileNotFoundError(f"checkpoint path {self.checkpoint_path} doesn't exist")
        files = sorted(glob.glob(os.path.join(self.checkpoint_path, "*.avro")))
        if len(files) != 1:
            if catch_exception:
                return None
            raise ValueError(f"Load model failed, no model file or multiple model files found in the model directory {self.checkpoint_path}")
        D = self.num_features
        ic = 1 if self.has_intercept else 0
        theta = self._load_model_native(files[0], D, ic)
        if theta is not None:
            return theta
        from .io.features import get_feature_map
        fmap = get_feature_map(self.feature_file) if self.feature_file else {}
        rec = next(iter(avro.read_file(files[0])))
        theta = np.zeros(D + ic)
        for m in rec["means"]:
            if m["name"] == constants.INTERCEPT and m["term"] == "":
                if ic:
                    theta[D] = m["value"]
            else:
                j = fmap.get((m["name"], m["term"]))
                if j is not None and j < D:
                    theta[j] = m["value"]
        return theta

    def _load_model_native(self, path, D, ic):
        """The same coefficients through libgdmix_io.so (a million (name, term, value) triples decode in milliseconds instead
        of seconds); None when the file is not of the plain layout this trainer and photon-ml write — intercept first, every
        feature in the feature file — and the record-by-record Python decoder has to apply the reference's lenient rules."""
        if not (self.feature_file and ic and native_reader.available()):
            return None
        try:
            schema, codec, sync, data_offset = avro.read_header(path)
            if codec not in ("null", "deflate") or not avro.is_model_schema(schema):
                return None
            prefix = self._feature_prefixes()[1]
            m = native_reader.read_models_avro(path, data_offset, sync, codec == "deflate", prefix,
                                               avro.enc_string(constants.INTERCEPT) + avro.enc_string(""), True)
        except (KeyError, AssertionError, ValueError):
            return None
        if len(m["ids"]) < 1:
            return None
        c0, c1 = int(m["coef_ptr"][0]), int(m["coef_ptr"][1])       # the first record, as next(iter(...)) takes
        idx = m["feat_idx"][c0:c1 - 1]
        if idx.size and int(idx.max()) >= D:
            return None
        theta = np.zeros(D + ic)
        theta[D] = m["mean"][c0]
        theta[idx] = m["mean"][c0 + 1:c1]      # of a feature listed twice the later value, as the Python loop leaves it
        return theta

    def export(self, output_model_dir):
        logger.info("No need model export for LR model.")

    # ---- predict ----------------------------------------------------------------------------------------------------
    def predict(self, output_dir, input_data_path, metadata_file, checkpoint_path, execution_context, schema_params):
        task_index = execution_context[constants.TASK_INDEX]
        num_workers = execution_context[constants.NUM_WORKERS]
        data = self._read(input_data_path, num_workers, task_index, schema_params)
        theta = self._load_model()
        self._score_and_write(theta, data, task_index, schema_params, output_dir)
