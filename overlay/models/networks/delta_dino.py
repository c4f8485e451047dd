"""Drop-in for the reference's models/networks/delta_dino.py."""
from dino_tracker_amd.networks import DeltaDINO  # noqa: F401
