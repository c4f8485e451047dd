"""TEST INFRASTRUCTURE ONLY -- minimal stand-in for torchvision (absent here) so the un-modified
reference modules import (data/data_utils.py:8, utils.py:4). Only ToTensor/Normalize are functional."""
from . import transforms  # noqa: F401
