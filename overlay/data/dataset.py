"""Drop-in for the reference's data/dataset.py: RangeNormalizer (hot-path glue) and the trajectory samplers of the
per-video test-time training (dino_tracker.py:78-87)."""
from dino_tracker_amd.dataset import DinoTrackerSampler, LongRangeSampler, RangeNormalizer  # noqa: F401
