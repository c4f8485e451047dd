"""Drop-in for the reference's dino_tracker.py: the reference's own `DINOTracker` (configuration, paths, model / optimizer /
scheduler set-up, checkpoints, logging -- loaded from the checkout, which must be importable further down `sys.path` or named
by $DTK_REFERENCE_ROOT) with the training iteration and its loss terms replaced by dino_tracker_amd/trainer.py.
DTK_TRAINER=reference keeps the inherited loop."""
import importlib.util
import os
import sys


def _reference_file():
    here = os.path.dirname(os.path.abspath(__file__))
    for d in [os.environ.get("DTK_REFERENCE_ROOT")] + list(sys.path):
        if d is None:
            continue
        f = os.path.join(os.path.abspath(d or "."), "dino_tracker.py")
        if os.path.isfile(f) and os.path.dirname(f) != here:
            return f
    raise ImportError("overlay/dino_tracker.py: the reference checkout's dino_tracker.py was not found on sys.path "
                      "(put the checkout on PYTHONPATH after this directory, or set DTK_REFERENCE_ROOT)")


_spec = importlib.util.spec_from_file_location("_dtk_reference_dino_tracker", _reference_file())
reference = importlib.util.module_from_spec(_spec)
sys.modules[_spec.name] = reference
_spec.loader.exec_module(reference)

from dino_tracker_amd.trainer import make_trainer  # noqa: E402

device = reference.device
DINOTracker = make_trainer(reference.DINOTracker)
