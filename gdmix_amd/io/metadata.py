"""Tensor metadata JSON (gdmix-trainer/src/gdmix/io/dataset_metadata.py:5-130): features[] / labels[] of
{name, dtype in {int,long,float,double,bytes,string}, shape, isSparse}."""
import json
import os
from collections import namedtuple

MetadataInfo = namedtuple("MetadataInfo", ["name", "dtype", "shape", "isSparse"])
SUPPORTED_TYPES = frozenset(["int", "long", "float", "double", "bytes", "string"])
INT_TYPES = frozenset(["int", "long"])


def read_json_file(path):
    if not os.path.exists(path):
        raise IOError(f"Path {path!r} does not exist.")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception as e:
        raise ValueError(f"Failed loading file {path!r}.") from e


class DatasetMetadata:
    FEATURES = "features"
    LABELS = "labels"
    INDICES = "indices"
    VALUES = "values"

    def __init__(self, path_or_metadata):
        md = read_json_file(path_or_metadata) if isinstance(path_or_metadata, str) else path_or_metadata
        if not isinstance(md.get(self.FEATURES, []), list):
            raise TypeError(f"Features must be a list. Type {type(md[self.FEATURES])} detected.")
        if not isinstance(md.get(self.LABELS, []), list):
            raise TypeError(f"Labels must be a list. Type {type(md[self.LABELS])} detected.")

        def parse(key):
            tensors = {}
            for ent in md.get(key, []):
                name = ent.get("name")
                if name in tensors:
                    raise ValueError(f"Invalid field: Tensor name in your metadata appears more than once:{name}")
                tensors[name] = self._build(ent)
            return tensors
        feats, labels = parse(self.FEATURES), parse(self.LABELS)
        self._tensors = {**feats, **labels}
        self._features = list(feats.values())
        self._labels = list(labels.values())

    @staticmethod
    def _build(d):
        if not {"name", "dtype", "shape"}.issubset(d.keys()):
            raise ValueError(f"Invalid field: required metadata fields are name,dtype,shape,isSparse; got {sorted(d)}")
        name, dtype, shape = d["name"], d["dtype"], d["shape"]
        if name is None or not isinstance(name, str):
            raise ValueError("Invalid field: Feature name can not be None and must be str")
        if dtype not in SUPPORTED_TYPES:
            raise ValueError(f"Invalid field: User provided dtype '{dtype}' is not supported. "
                             f"Supported types are '{sorted(SUPPORTED_TYPES)}'.")
        if shape is None or not isinstance(shape, list):
            raise ValueError("Invalid field: Feature shape can not be None and must be a list")
        return MetadataInfo(name, dtype, shape, bool(d.get("isSparse", False)))

    def get_features(self):
        return list(self._features)

    def get_labels(self):
        return list(self._labels)

    def get_feature_names(self):
        return [t.name for t in self._features]

    def get_label_names(self):
        return [t.name for t in self._labels]

    def get_feature_shape(self, feature_name):
        return next(t for t in self._features if t.name == feature_name).shape

    def get_tensors(self):
        return dict(self._tensors)
