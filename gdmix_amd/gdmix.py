"""CLI entry of the random-effect stage, flag-compatible with the reference's
`python -m gdmix.gdmix --stage=random_effect ...` (gdmix-trainer/src/gdmix/gdmix.py:13-36): one flat argv is
shared by Params, SchemaParams and REParams, unknown flags are ignored, any failure exits non-zero.

    python -m gdmix_amd.gdmix --stage=random_effect --action=train --model_type=logistic_regression \\
        --partition_list_file=... --training_data_dir=... --metadata_file=... --output_model_dir=... ...

--stage=fixed_effect runs the linear / logistic fixed-effect model (fe_model.py); the DeText stage is out of scope.
"""
import logging
import sys


def process_defaults():
    """Settings of the PROCESS, made by its entry points (this CLI, bench.py) before the first device call — never by importing the
    package, which an embedding host may do with a runtime it configured itself. GPU_MAX_HW_QUEUES: a context deals its size classes
    over four streams (gdmix_re_set_spread) and the host pipeline keeps three contexts busy; twelve streams on the four hardware
    queues a process gets by default alias onto each other (hand-over 72 -> 79 M entities/s with eight, tools/ab.sh hwq). Read by
    the HIP runtime when it initialises; a value set by the caller wins."""
    import os
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


if __name__ == "__main__":      # (before the imports below can load the HIP runtime)
    process_defaults()

from . import constants  # noqa: E402
from .driver import FixedEffectDriver, RandomEffectDriver
from .model import RandomEffectLRLBFGSModel
from .params import Params, SchemaParams

logging.basicConfig(level=logging.INFO)
logger = logging.getLogger(__name__)


def run(args):
    params = Params.__from_argv__(args, error_on_unknown=False)
    schema_params = SchemaParams.__from_argv__(args, error_on_unknown=False)
    logger.info(f"Parsed schema params amd gdmix args (params): {params}")
    if params.stage == constants.FIXED_EFFECT:
        if params.model_type not in (constants.LOGISTIC_REGRESSION, constants.LINEAR_REGRESSION):
            raise NotImplementedError(f"model type {params.model_type!r}: the fixed effect runs logistic_regression and linear_regression")
        from .fe_model import FixedEffectLRModelLBFGS
        driver = FixedEffectDriver(base_training_params=params, model=FixedEffectLRModelLBFGS(raw_model_params=args, base_training_params=params))
    elif params.stage == constants.RANDOM_EFFECT:
        if params.model_type != constants.LOGISTIC_REGRESSION:
            raise ValueError("Random effect supports logistic_regression only")
        driver = RandomEffectDriver(base_training_params=params, model=RandomEffectLRLBFGSModel(raw_model_params=args))
    else:
        raise NotImplementedError(f"stage {params.stage!r} does not run on this library")
    if params.action == constants.ACTION_TRAIN:
        driver.run_training(schema_params=schema_params, export_model=True)
    elif params.action == constants.ACTION_INFERENCE:
        driver.run_inference(schema_params=schema_params)
    else:
        raise Exception(f"Unsupported action {params.action}")


if __name__ == "__main__":
    run(sys.argv)
