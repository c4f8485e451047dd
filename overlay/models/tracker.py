"""Drop-in for the reference's models/tracker.py: re-exports the HIP-backed Tracker."""
from dino_tracker_amd.tracker import EPS, Tracker, load_pre_trained_model  # noqa: F401
