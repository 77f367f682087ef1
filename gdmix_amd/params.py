"""Parameter classes of the random-effect stage, field for field as the reference declares them:
GDMixParams / SchemaParams / Params (gdmix-trainer/src/gdmix/params.py:12-50), LRParams
(models/custom/base_lr_params.py:5-42) and REParams (models/custom/random_effect_lr_lbfgs_model.py:34-53).
"""
from dataclasses import dataclass
from typing import ClassVar, Optional

from . import constants
from .argv import from_argv, to_argv

_ACTIONS = (constants.ACTION_INFERENCE, constants.ACTION_TRAIN)
_STAGES = (constants.FIXED_EFFECT, constants.RANDOM_EFFECT)
_MODEL_TYPES = (constants.LOGISTIC_REGRESSION, constants.LINEAR_REGRESSION, constants.DETEXT)
_VARIANCE_MODE = (constants.FULL, constants.SIMPLE)


class _ArgvMixin:
    @classmethod
    def __from_argv__(cls, argv, error_on_unknown=False):
        return from_argv(cls, argv, error_on_unknown=error_on_unknown)

    def __to_argv__(self):
        return to_argv(self)


@dataclass
class SchemaParams(_ArgvMixin):
    uid_column_name: str                                    # Unique id column name in the train/validation data.
    weight_column_name: Optional[str] = None                # weight column name in the train/validation data.
    label_column_name: Optional[str] = None                 # Label column name in the train/validation data.
    prediction_score_column_name: Optional[str] = None      # Prediction score column name in the generated result file.
    prediction_score_per_coordinate_column_name: str = "predictionScorePerCoordinate"


@dataclass
class Params(_ArgvMixin):
    """GDMix driver parameters = SchemaParams + GDMixParams (params.py:12-50)."""
    uid_column_name: str
    weight_column_name: Optional[str] = None
    label_column_name: Optional[str] = None
    prediction_score_column_name: Optional[str] = None
    prediction_score_per_coordinate_column_name: str = "predictionScorePerCoordinate"
    action: str = constants.ACTION_TRAIN
    stage: str = constants.FIXED_EFFECT
    model_type: str = constants.LOGISTIC_REGRESSION
    training_score_dir: Optional[str] = None
    validation_score_dir: Optional[str] = None
    partition_list_file: Optional[str] = None

    def __post_init__(self):
        assert self.action in _ACTIONS, f"Action: {self.action} must be in {_ACTIONS}"
        assert self.stage in _STAGES, f"Stage: {self.stage} must be in {_STAGES}"
        assert self.model_type in _MODEL_TYPES, f"Model type: {self.model_type} must be in {_MODEL_TYPES}"
        assert (self.action == constants.ACTION_TRAIN and self.label_column_name) or \
               (self.action == constants.ACTION_INFERENCE and self.prediction_score_column_name)


@dataclass
class LRParams(_ArgvMixin):
    """Base linear model parameters (base_lr_params.py:5-42)."""
    metadata_file: str
    output_model_dir: str
    training_data_dir: Optional[str] = None
    validation_data_dir: Optional[str] = None
    feature_bag: Optional[str] = None
    feature_file: Optional[str] = None
    regularize_bias: bool = True
    l2_reg_weight: float = 1.0
    lbfgs_tolerance: float = 1e-12
    num_of_lbfgs_curvature_pairs: int = 10
    num_of_lbfgs_iterations: int = 100
    has_intercept: bool = True
    offset_column_name: str = "offset"
    batch_size: int = 16
    data_format: str = "tfrecord"
    # un-annotated in the reference, hence not settable from argv (base_lr_params.py:32)
    sparsity_threshold: ClassVar[float] = 1.0e-4

    def __post_init__(self):
        assert self.batch_size > 0, "Batch size must be positive number"
        if self.regularize_bias:
            assert self.has_intercept, "Intercept must be used when it is regularized"
        assert self.feature_bag or self.has_intercept, "Either intercept or feature bag much be used"


@dataclass
class REParams(LRParams):
    """Random-effect model parameters (random_effect_lr_lbfgs_model.py:34-53). The queue / consumer knobs
    are accepted for CLI compatibility; the device solver has no job queue and ignores them."""
    partition_entity: Optional[str] = None
    enable_local_indexing: bool = False
    max_training_queue_size: int = 10
    training_queue_timeout_in_seconds: int = 300
    num_of_consumers: int = 2
    random_effect_variance_mode: Optional[str] = None
    disable_random_effect_scoring_after_training: bool = False
    # not in the reference: move entities between the workers of a node when partitions are skewed (rebalance.py)
    rebalance_entities: bool = False

    def __post_init__(self):
        # the reference's REParams.__post_init__ does NOT chain to LRParams.__post_init__ (random_effect_lr_lbfgs_model.py:
        # 48-53): `--has_intercept False` with regularize_bias left at its default True is a valid random-effect
        # configuration upstream (test_random_effect_lr_lbfgs_model.py: warm start without intercept)
        assert self.max_training_queue_size > self.num_of_consumers, \
            "queue size limit must be larger than the number of consumers"
        assert self.random_effect_variance_mode is None or self.random_effect_variance_mode in _VARIANCE_MODE, \
            f"Action: {self.random_effect_variance_mode} must be in {_VARIANCE_MODE}"
