"""Feature list file: CSV, one `name,term` per line, index = zero-based line number
(gdmix-trainer/src/gdmix/util/io_utils.py:215-239)."""
import csv


def read_feature_list(feature_file):
    result = []
    with open(feature_file, newline="") as f:
        for row in csv.reader(f):
            assert len(row) == 2, f"Each feature name should have exactly name and term only, but I got {row}."
            result.append(tuple(row))
    return result


def get_feature_map(feature_file):
    return {feature: index for index, feature in enumerate(read_feature_list(feature_file))}
