"""The strings the CLI, the drivers and the model classes exchange: stage / action / model-type names, directory names
of the partitioner's layout, the keys of the execution context. They are the reference's values (its command lines,
directory layout and model files must keep working unchanged: gdmix-trainer/src/gdmix/util/constants.py), grouped here
by what uses them."""

# --stage / --action / --model_type
ACTION_TRAIN, ACTION_INFERENCE = "train", "inference"
FIXED_EFFECT, RANDOM_EFFECT, DETEXT = "fixed_effect", "random_effect", "detext"
LOGISTIC_REGRESSION, LINEAR_REGRESSION = "logistic_regression", "linear_regression"

# data layout: <training_data_dir>/{active,passive}/partitionId=K/*.tfrecord; variance modes; the intercept's feature name
ACTIVE, PASSIVE = "active", "passive"
TFRECORD, TFRECORD_GLOB_PATTERN = "tfrecord", "*.tfrecord"
SIMPLE, FULL = "simple", "full"
INTERCEPT = "(INTERCEPT)"

# execution context handed from a driver to a model (drivers/driver.py:191-216, drivers/random_effect_driver.py:28-58)
(PARTITION_INDEX, TASK_TYPE, TASK_INDEX, CLUSTER_SPEC, NUM_WORKERS, NUM_SHARDS, SHARD_INDEX, IS_CHIEF) = (
    "partition_index", "task_type", "task_index", "cluster_spec", "num_workers", "num_shards", "shard_index", "is_chief")
WORKER = TASK_TYPE_WORKER = "worker"
TF_CONFIG = "TF_CONFIG"
(ACTIVE_TRAINING_OUTPUT_FILE, PASSIVE_TRAINING_OUTPUT_FILE, VALIDATION_OUTPUT_FILE, PASSIVE_TRAINING_DATA_DIR) = (
    "active_training_output_file", "passive_training_output_file", "validation_output_file", "passive_training_data_dir")

# modelClass written into the photon-ml model records
PHOTON_LR_MODEL_CLASS = "com.linkedin.photon.ml.supervised.classification.LogisticRegressionModel"
