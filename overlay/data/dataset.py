"""Drop-in for the reference's data/dataset.py: RangeNormalizer (hot-path glue).  The trajectory samplers are
training-only and are not provided by this implementation."""
from dino_tracker_amd.dataset import RangeNormalizer  # noqa: F401


class _TrainingOnly:
    def __init__(self, *a, **k):
        raise NotImplementedError("test-time training (dino_tracker.py:392-448) is outside the inference hot path")


class LongRangeSampler(_TrainingOnly):
    pass


class DinoTrackerSampler(_TrainingOnly):
    pass
