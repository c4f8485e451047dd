"""CPU: the overlay makes the reference's module paths resolve to this implementation (namespace-package shadowing),
while non-hot-path modules still come from the reference checkout when it is present."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
import models.tracker, models.model_inference, models.extractor, data.dataset, utils
import models.networks.tracker_head, models.networks.delta_dino, models.networks.conv_norm
import dino_tracker_amd.tracker as T, dino_tracker_amd.model_inference as MI, dino_tracker_amd.extractor as E
assert models.tracker.Tracker is T.Tracker
assert models.model_inference.ModelInference is MI.ModelInference
assert models.extractor.VitExtractor is E.VitExtractor
assert data.dataset.RangeNormalizer(shapes=(10, 10, 3)).normalizer.device.type == "cpu"
assert callable(utils.add_config_paths) and callable(utils.get_dino_features_video)
for m in (models.tracker, models.model_inference, data.dataset, utils):
    assert "overlay" in m.__file__, m.__file__
if REF:
    import models.utils, data.tapvid
    assert REF in models.utils.__file__ and REF in data.tapvid.__file__
    import dino_tracker                       # the reference's orchestrator imports cleanly on top of the overlay
    assert dino_tracker.Tracker is T.Tracker
print("overlay ok")
'''


def test_overlay_resolution():
    ref = "/root/reference" if os.path.isdir("/root/reference/models") else ""
    paths = [os.path.join(ROOT, "overlay"), ROOT] + ([os.path.join(ROOT, "oracle", "shims"), ref] if ref else [])
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(paths))
    r = subprocess.run([sys.executable, "-c", f"REF = {ref!r}\n" + SCRIPT], capture_output=True, text=True, env=env,
                       cwd="/tmp", timeout=300)
    assert r.returncode == 0 and "overlay ok" in r.stdout, r.stdout + r.stderr
