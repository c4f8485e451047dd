"""TEST INFRASTRUCTURE ONLY -- imports the UN-MODIFIED reference from /root/reference on CPU.

Nothing in the product (`dino_tracker_amd/`) may import this module.  It exists so that
  * `tests/golden/make_golden.py` can run the real reference here and write golden fixtures, and
  * CPU tests in THIS container can pin `oracle/ref_algo.py` against the real reference.
`/root/reference` does not exist on the GPU box, so everything here is guarded by `available()`.

How: the reference has no `__init__.py`, so `models`, `data`, ... are namespace
packages; we put `oracle/shims` (stubs for antialiased_cnns / torchvision.transforms / cv2 / imageio)
and `/root/reference` on sys.path, and patch the one CPU-hostile default:
`data.dataset.RangeNormalizer.__init__(..., device='cuda')` (/root/reference/data/dataset.py:15).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DTK_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")
_loaded = None


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "tracker.py"))


def load() -> types.SimpleNamespace:
    """Returns a namespace with the reference modules (tracker, model_inference, tracker_head,
    delta_dino, models_utils, utils, dataset, extractor)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    for p in (REFERENCE_ROOT, _SHIMS):
        if p in sys.path:
            sys.path.remove(p)
    # shims first so the stubs win over anything site-wide; reference second
    sys.path.insert(0, REFERENCE_ROOT)
    sys.path.insert(0, _SHIMS)
    for name in ("models", "data", "utils"):
        mod = sys.modules.get(name)
        if mod is not None and REFERENCE_ROOT not in str(getattr(mod, "__path__", getattr(mod, "__file__", ""))):
            raise RuntimeError(f"module {name!r} already imported from elsewhere: {mod}")
    import data.dataset as dataset  # noqa: E402
    dataset.RangeNormalizer.__init__.__defaults__ = ("cpu",)
    import utils as ref_utils  # noqa: E402
    import models.utils as models_utils  # noqa: E402
    import models.networks.tracker_head as tracker_head  # noqa: E402
    import models.networks.delta_dino as delta_dino  # noqa: E402
    import models.tracker as tracker  # noqa: E402
    import models.model_inference as model_inference  # noqa: E402
    import models.extractor as extractor  # noqa: E402
    _loaded = types.SimpleNamespace(
        dataset=dataset, utils=ref_utils, models_utils=models_utils, tracker_head=tracker_head,
        delta_dino=delta_dino, tracker=tracker, model_inference=model_inference, extractor=extractor)
    return _loaded
