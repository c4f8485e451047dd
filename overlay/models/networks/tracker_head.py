"""Drop-in for the reference's models/networks/tracker_head.py."""
from dino_tracker_amd.networks import TrackerHead  # noqa: F401
