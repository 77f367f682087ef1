"""RandomEffectDriver: partition list -> per-worker partition striding -> model.train / model.predict per
partition, with the directory and file naming of the reference
(gdmix-trainer/src/gdmix/drivers/driver.py:85-216, drivers/random_effect_driver.py:12-73).

One process per GPU: the worker identity comes from TF_CONFIG exactly as in the reference; when it is absent
and the process was started by torch.distributed.run, RANK / WORLD_SIZE are used instead, so that
`python -m torch.distributed.run --nproc-per-node 8 -m gdmix_amd.gdmix ...` shards the partitions over the 8
GPUs of a node (worker w trains partitions[w::num_workers], random_effect_driver.py:60-68).
"""
import glob
import json
import logging
import os

from . import constants

logger = logging.getLogger(__name__)
logger.setLevel(logging.INFO)


# partitions decoded ahead of the one being solved (3 with 12 threads per native call against 2 x 32: warm-started run 0.51 -> 0.44 s per
# million entities, cold 0.27 -> 0.24; tools/r04_e2e_knobs.sh)
PREFETCH_PARTITIONS = int(os.environ.get("GDMIX_PREFETCH_PARTITIONS", "3"))


def is_empty_directory(input_dir):
    if not os.path.isdir(input_dir):
        raise ValueError(f"Directory expected, but {input_dir} is not a directory")
    return len(os.listdir(input_dir)) == 0


class RandomEffectDriver:
    _RANDOM_EFFECT_PARTITION_DIR_PREFIX = "partitionId="

    def __init__(self, base_training_params, model):
        self.base_training_params = base_training_params
        self.model = model
        self._validate_params()
        self.execution_context = self._setup_cluster()
        self.effect_name = constants.RANDOM_EFFECT

    def _validate_params(self):
        assert self.base_training_params.model_type == constants.LOGISTIC_REGRESSION, \
            "Random effect supports logistic_regression only"
        assert self.base_training_params.partition_list_file is not None, \
            "Random effect requires partition list file"

    def _setup_cluster(self):
        tf_config = os.environ.get(constants.TF_CONFIG)
        if not tf_config:
            rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
            return {constants.TASK_TYPE: "worker", constants.TASK_INDEX: rank, constants.CLUSTER_SPEC: None,
                    constants.NUM_WORKERS: world, constants.NUM_SHARDS: 1, constants.SHARD_INDEX: 0,
                    constants.IS_CHIEF: rank == 0}
        cfg = json.loads(tf_config)
        cluster = cfg.get("cluster") or {}
        task = cfg.get("task", {})
        ctx = {constants.TASK_TYPE: task.get("type"), constants.TASK_INDEX: task.get("index"),
               constants.CLUSTER_SPEC: None,   # random effect runs in local mode
               constants.NUM_WORKERS: len(cluster.get(constants.WORKER, [])),
               constants.NUM_SHARDS: 1, constants.SHARD_INDEX: 0, constants.IS_CHIEF: task.get("index") == 0}
        if ctx[constants.TASK_TYPE] is None or ctx[constants.TASK_INDEX] is None:
            raise Exception("No job name found")
        if ctx[constants.NUM_WORKERS] < 1:
            raise Exception("No worker found")
        os.environ.pop(constants.TF_CONFIG, None)   # random effect runs in local mode
        return ctx

    def _get_partition_list(self):
        with open(self.base_training_params.partition_list_file) as f:
            line = f.readline()
        all_partitions = [int(x) for x in line.split(",")]
        idx = range(self.execution_context[constants.TASK_INDEX], len(all_partitions),
                    self.execution_context[constants.NUM_WORKERS])
        return [all_partitions[i] for i in idx]

    def _init_collectives(self):
        """Entity re-balancing exchanges entities between the workers: RCCL when launched one process per GPU by
        torch.distributed.run (which provides the rendezvous variables; TF_CONFIG alone does not)."""
        if self.execution_context[constants.NUM_WORKERS] <= 1:
            return
        import torch
        import torch.distributed as dist
        if dist.is_initialized():
            return
        if "RANK" not in os.environ or "MASTER_ADDR" not in os.environ:
            raise RuntimeError("--rebalance_entities needs the workers to be started by torch.distributed.run "
                               "(RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT): they exchange entities over RCCL")
        backend = "nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", "1")) else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(os.environ.get("GDMIX_DIST_BACKEND", backend))

    def _anchor_directory(self, directory_path, partition_index):
        return os.path.join(directory_path, self._RANDOM_EFFECT_PARTITION_DIR_PREFIX + str(partition_index))

    def run_training(self, schema_params, export_model=False, output_model_dir=None):
        logger.info(f"Commencing {self.effect_name} training")
        logger.info(f"Execution context : {self.execution_context}")
        partition_index_list = self._get_partition_list()
        logger.info(f"This worker on work on the following list of partitions : {partition_index_list}")
        # With entity re-balancing the workers train in lockstep, one partition each per round; a worker that has no
        # partition (or an empty one) in a round still joins that round's collectives.
        lockstep = bool(getattr(getattr(self.model, "model_params", None), "rebalance_entities", False))
        rounds = len(partition_index_list)
        if lockstep:
            self._init_collectives()
            with open(self.base_training_params.partition_list_file) as f:
                total = len(f.readline().split(","))
            w = self.execution_context[constants.NUM_WORKERS]
            rounds = (total + w - 1) // w
        pipelined = hasattr(self.model, "begin_pipeline")
        if pipelined:
            self.model.begin_pipeline()
        try:
            self._train_rounds(rounds, partition_index_list, lockstep, schema_params, export_model, output_model_dir, pipelined)
        finally:
            if pipelined:
                self.model.end_pipeline()

    def _plan_group(self, k, partition_index_list, schema_params):
        """From round k on: the consecutive partitions without a prior model, up to the model's limits (count, bytes of input), are handed
        to the model as one group — it solves them in one device batch and still writes every partition's own files (model.plan_group).
        -> the number of rounds the plan covers (1: partition k on its own)."""
        limit = getattr(self.model, "group_limits", None)
        if limit is None:
            return 1
        max_count, max_bytes = limit()
        out_dir = self.model.model_params.output_model_dir
        dirs, covered, total = [], 0, 0
        for p in partition_index_list[k:]:
            d = self._anchor_directory(self.model.training_data_dir, p)
            if is_empty_directory(d):        # skipped by the loop below, whatever group it falls into
                covered += 1
                continue
            if os.path.exists(os.path.join(out_dir, f"part-{p:05d}.avro")):    # warm start: on its own
                break
            size = sum(os.path.getsize(f) for f in glob.glob(os.path.join(d, "*")) if os.path.isfile(f))
            if dirs and (len(dirs) >= max_count or total + size > max_bytes):
                break
            dirs.append(d)
            total += size
            covered += 1
        if len(dirs) < 2 or self.model.plan_group(dirs, self.model.metadata_file, schema_params) != len(dirs):
            return 1
        return covered

    def _train_rounds(self, rounds, partition_index_list, lockstep, schema_params, export_model, output_model_dir, pipelined):
        planned_until = 0
        for k in range(rounds):
            if k >= len(partition_index_list):
                self.model.idle_round()
                continue
            partition_index = partition_index_list[k]
            if pipelined and not lockstep and k >= planned_until:
                planned_until = k + self._plan_group(k, partition_index_list, schema_params)
            if pipelined:   # decode the next partitions while this one is solved (prefetch() ignores what is already on its way)
                for ahead in range(1, PREFETCH_PARTITIONS + 1):
                    if k + ahead >= len(partition_index_list):
                        break
                    next_dir = self._anchor_directory(self.model.training_data_dir, partition_index_list[k + ahead])
                    if not is_empty_directory(next_dir):   # a partition that will be skipped is not decoded (nor kept) at all
                        self.model.prefetch(next_dir, self.model.metadata_file, schema_params)
                        if ahead == 1:
                            self.model.prefetch_prior_model(partition_index_list[k + 1])
            checkpoint_path = self._anchor_directory(self.model.checkpoint_path, partition_index)
            training_data_dir = self._anchor_directory(self.model.training_data_dir, partition_index)
            validation_data_dir = self._anchor_directory(self.model.validation_data_dir, partition_index) \
                if self.model.validation_data_dir else None
            if is_empty_directory(training_data_dir):
                logger.info(f"{training_data_dir} is empty, no dataset to train on.")
                if lockstep:
                    self.model.idle_round()
                continue
            self.execution_context[constants.PARTITION_INDEX] = partition_index
            if pipelined and validation_data_dir:   # decoded while the training data is solved
                self.model.prefetch(validation_data_dir, self.model.metadata_file, schema_params)
            self.model.train(training_data_dir=training_data_dir, validation_data_dir=validation_data_dir,
                             metadata_file=self.model.metadata_file, checkpoint_path=checkpoint_path,
                             execution_context=self._prepare_training_context(partition_index),
                             schema_params=schema_params)
            if export_model and self.execution_context[constants.IS_CHIEF]:
                self.model.export(output_model_dir=output_model_dir)

    def run_inference(self, schema_params):
        logger.info(f"Commencing {self.effect_name} inference")
        if self.execution_context[constants.TASK_TYPE] != constants.TASK_TYPE_WORKER:
            logger.info("Only workers should run inference. Exiting")
            return
        for partition_index in self._get_partition_list():
            self.execution_context[constants.PARTITION_INDEX] = partition_index
            for input_path, output_path in ((self.model.training_data_dir, self.base_training_params.training_score_dir),
                                            (self.model.validation_data_dir, self.base_training_params.validation_score_dir)):
                if input_path and output_path:
                    data_path = self._anchor_directory(input_path, partition_index)
                    output_dir = os.path.join(self._anchor_directory(output_path, partition_index))
                    if is_empty_directory(input_path):
                        logger.info(f"{input_path} is empty, no dataset to inference on.")
                        continue
                    self.model.predict(output_dir=output_dir, input_data_path=data_path,
                                       metadata_file=self.model.metadata_file, checkpoint_path=self.model.checkpoint_path,
                                       execution_context=self.execution_context, schema_params=schema_params)
        logger.info("Inference complete")

    def _prepare_training_context(self, partition_index):
        p = self.base_training_params
        task = self.execution_context[constants.TASK_INDEX]
        anchored = self._anchor_directory(p.training_score_dir, partition_index)
        ctx = dict(self.execution_context)
        passive = self._anchor_directory(self.model.passive_training_data_dir, partition_index)
        if os.path.exists(passive) and len(glob.glob(os.path.join(passive, "[!.]*"))) != 0:
            ctx[constants.PASSIVE_TRAINING_DATA_DIR] = passive
        ctx[constants.ACTIVE_TRAINING_OUTPUT_FILE] = os.path.join(anchored, f"part-{task:05d}-active.avro")
        ctx[constants.PASSIVE_TRAINING_OUTPUT_FILE] = os.path.join(anchored, f"part-{task:05d}-passive.avro")
        ctx[constants.VALIDATION_OUTPUT_FILE] = os.path.join(
            self._anchor_directory(p.validation_score_dir, partition_index), f"part-{task:05d}.avro") \
            if p.validation_score_dir else None
        return ctx


class FixedEffectDriver:
    """drivers/fixed_effect_driver.py:12-75 + drivers/driver.py:85-216: one model over all the data; worker w of W reads
    files[w::W]; the partition index is the task index and directories are not anchored."""

    def __init__(self, base_training_params, model):
        self.base_training_params = base_training_params
        self.model = model
        self.execution_context = self._setup_cluster()
        self.effect_name = constants.FIXED_EFFECT

    def _setup_cluster(self):
        tf_config = os.environ.get(constants.TF_CONFIG)
        if not tf_config:
            rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
            return {constants.TASK_TYPE: "worker", constants.TASK_INDEX: rank, constants.CLUSTER_SPEC: None,
                    constants.NUM_WORKERS: world, constants.NUM_SHARDS: world, constants.SHARD_INDEX: rank,
                    constants.IS_CHIEF: rank == 0}
        cfg = json.loads(tf_config)
        cluster = cfg.get("cluster") or {}
        task = cfg.get("task", {})
        n = len(cluster.get(constants.WORKER, []))
        ctx = {constants.TASK_TYPE: task.get("type"), constants.TASK_INDEX: task.get("index"), constants.CLUSTER_SPEC: None,
               constants.NUM_WORKERS: n, constants.NUM_SHARDS: n, constants.SHARD_INDEX: task.get("index"),
               constants.IS_CHIEF: task.get("index") == 0}
        if ctx[constants.TASK_TYPE] is None or ctx[constants.TASK_INDEX] is None:
            raise Exception("No job name found")
        if n < 1:
            raise Exception("No worker found")
        return ctx

    def _get_partition_list(self):
        return [self.execution_context[constants.TASK_INDEX]]

    def _init_collectives(self):
        """W > 1 workers all-reduce gradient and value: RCCL when launched one process per GPU by torch.distributed.run."""
        if self.execution_context[constants.NUM_WORKERS] <= 1:
            return
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            backend = "nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", "1")) else "gloo"
            dist.init_process_group(os.environ.get("GDMIX_DIST_BACKEND", backend))

    def run_training(self, schema_params, export_model=False, output_model_dir=None):
        logger.info(f"Commencing {self.effect_name} training")
        self._init_collectives()
        ctx = dict(self.execution_context)
        ctx[constants.PARTITION_INDEX] = ctx[constants.TASK_INDEX]
        self.model.train(training_data_dir=self.model.training_data_dir, validation_data_dir=self.model.validation_data_dir,
                         metadata_file=self.model.metadata_file, checkpoint_path=self.model.checkpoint_path,
                         execution_context=ctx, schema_params=schema_params)
        if export_model and ctx[constants.IS_CHIEF]:
            self.model.export(output_model_dir=output_model_dir)

    def run_inference(self, schema_params):
        logger.info(f"Commencing {self.effect_name} inference")
        ctx = dict(self.execution_context)
        ctx[constants.PARTITION_INDEX] = ctx[constants.TASK_INDEX]
        self.model.predict(output_dir=self.base_training_params.validation_score_dir,
                           input_data_path=self.model.validation_data_dir, metadata_file=self.model.metadata_file,
                           checkpoint_path=self.model.checkpoint_path, execution_context=ctx, schema_params=schema_params)
