"""Drop-in for the reference's models/extractor.py."""
from dino_tracker_amd.extractor import VitExtractor  # noqa: F401
