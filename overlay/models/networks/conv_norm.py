"""Drop-in for the reference's models/networks/conv_norm.py."""
from dino_tracker_amd.networks import NormalizedConv2d  # noqa: F401
