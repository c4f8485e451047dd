"""CPU: the overlay makes the reference's module paths resolve to this implementation (namespace-package shadowing),
while non-hot-path modules still come from the reference checkout when it is present -- and it does so for a REAL
script file sitting in a reference-shaped directory, which is the case `python script.py` + PYTHONPATH gets wrong
(the script's directory precedes PYTHONPATH): that is what `python -m dino_tracker_amd.run` is for."""
import os

import pytest
import subprocess
import sys
import textwrap

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
    import dino_tracker                       # overlay/dino_tracker.py: the reference's trainer class, loaded from the
    assert dino_tracker.reference.Tracker is T.Tracker                  # checkout on top of the overlay's models, ...
    assert issubclass(dino_tracker.DINOTracker, dino_tracker.reference.DINOTracker)   # ... with the device-side iteration
    assert dino_tracker.DINOTracker.train is not dino_tracker.reference.DINOTracker.train
    assert REF in dino_tracker.reference.__file__
print("overlay ok")
'''


def test_overlay_resolution():
    ref = "/root/reference" if os.path.isdir("/root/reference/models") else ""
    paths = [os.path.join(ROOT, "overlay"), ROOT] + ([os.path.join(ROOT, "oracle", "shims"), ref] if ref else [])
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(paths))
    r = subprocess.run([sys.executable, "-c", f"REF = {ref!r}\n" + SCRIPT], capture_output=True, text=True, env=env,
                       cwd="/tmp", timeout=300)
    assert r.returncode == 0 and "overlay ok" in r.stdout, r.stdout + r.stderr


def _fake_reference(tmp_path):
    """A directory shaped like the reference checkout: a script in its root next to `models/`, `data/`, `utils.py`
    whose hot-path modules are decoys (importing one of them is the failure)."""
    root = tmp_path / "dino-tracker"
    for d in ("models", "models/networks", "data"):
        (root / d).mkdir(parents=True)
    decoy = 'raise ImportError("decoy: the reference module was imported instead of the overlay")\n'
    for f in ("models/tracker.py", "models/model_inference.py", "models/extractor.py", "models/networks/tracker_head.py",
              "data/dataset.py", "utils.py"):
        (root / f).write_text(decoy)
    (root / "models" / "utils.py").write_text("WHO = 'reference models/utils.py'\n")  # not shadowed by the overlay
    (root / "probe.py").write_text(textwrap.dedent('''
        import sys
        from models.tracker import Tracker
        from models.model_inference import ModelInference
        from data.dataset import RangeNormalizer
        import utils, models.utils
        assert __name__ == "__main__" and sys.argv[1:] == ["--flag", "7"], sys.argv
        assert models.utils.WHO == "reference models/utils.py"
        print("probe ok", Tracker.__module__, utils.__file__)
    '''))
    return root


def test_launcher_runs_a_script_from_the_reference_root(tmp_path):
    root = _fake_reference(tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "dino_tracker_amd.run", str(root / "probe.py"), "--flag", "7"],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "probe ok dino_tracker_amd.tracker" in r.stdout and os.path.join("overlay", "utils.py") in r.stdout


def test_plain_python_with_pythonpath_hits_the_reference_modules(tmp_path):
    """The failure mode the launcher exists for: documented, so INTEGRATION.md cannot regress to it."""
    root = _fake_reference(tmp_path)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "overlay"), ROOT]))
    r = subprocess.run([sys.executable, str(root / "probe.py"), "--flag", "7"], capture_output=True, text=True, env=env,
                       cwd=str(tmp_path), timeout=300)
    assert r.returncode != 0 and "decoy" in r.stderr


def test_launcher_fails_loudly_when_a_hot_module_is_not_the_overlay(tmp_path):
    root = _fake_reference(tmp_path)
    (root / "models" / "tracker.py").write_text("Tracker = None\n")
    (root / "sneaky.py").write_text("import sys, importlib.util\n"
                                    f"spec = importlib.util.spec_from_file_location('models.tracker', r'{root}/models/tracker.py')\n"
                                    "m = importlib.util.module_from_spec(spec); sys.modules['models.tracker'] = m\n")
    r = subprocess.run([sys.executable, "-m", "dino_tracker_amd.run", str(root / "sneaky.py")], capture_output=True,
                       text=True, env=dict(os.environ, PYTHONPATH=ROOT), cwd=str(tmp_path), timeout=300)
    assert r.returncode != 0 and "did not resolve to the overlay" in r.stderr


def test_run_videos_partitions_data_paths_over_ranks(tmp_path):
    """dino_tracker_amd.run_videos: rank r of a world-size-3 launch runs the script once per data path v = r (mod 3), each
    as a plain single-GPU process pinned to GPU LOCAL_RANK, with the script's own arguments passed through."""
    import json
    import subprocess
    import sys
    script = tmp_path / "fake_train.py"
    script.write_text("import argparse, json, os\n"
                      "p = argparse.ArgumentParser(); p.add_argument('--data-path'); p.add_argument('--config')\n"
                      "a = p.parse_args()\n"
                      "json.dump({'config': a.config, 'gpu': os.environ.get('HIP_VISIBLE_DEVICES'),\n"
                      "           'world': os.environ.get('WORLD_SIZE')}, open(os.path.join(a.data_path, 'done.json'), 'w'))\n")
    paths = []
    for i in range(7):
        d = tmp_path / f"video{i}"
        d.mkdir()
        paths.append(str(d))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rank in range(3):
        env = dict(os.environ, PYTHONPATH=root, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="3")
        r = subprocess.run([sys.executable, "-m", "dino_tracker_amd.run_videos", str(script), "--data-paths"] + paths +
                           ["--", "--config", "c.yaml"], env=env, capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-2000:]
    for i, d in enumerate(paths):
        rec = json.load(open(os.path.join(d, "done.json")))
        assert rec == {"config": "c.yaml", "gpu": str(i % 3), "world": None}, (i, rec)
    # a failing video makes its rank fail, the others still run
    bad = tmp_path / "bad.py"
    bad.write_text("import sys\nsys.exit(3 if 'video1' in sys.argv[2] else 0)\n")
    r = subprocess.run([sys.executable, "-m", "dino_tracker_amd.run_videos", str(bad), "--data-paths"] + paths[:3],
                       env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "video1" in r.stderr


def test_run_videos_pins_the_local_gpu_inside_an_existing_restriction():
    """a job already restricted to some GPUs (HIP_VISIBLE_DEVICES=4,5,6,7 from a scheduler) must hand rank r entry r
    of THAT list, not device r."""
    from dino_tracker_amd.run_videos import pin_local_gpu
    env = {"HIP_VISIBLE_DEVICES": "4,5,6,7"}
    pin_local_gpu(env, 2)
    assert env["HIP_VISIBLE_DEVICES"] == "6"
    env = {"CUDA_VISIBLE_DEVICES": "1, 3"}
    pin_local_gpu(env, 1)
    assert env == {"CUDA_VISIBLE_DEVICES": "3"}
    env = {"ROCR_VISIBLE_DEVICES": "2,3"}
    pin_local_gpu(env, 1)
    assert env["HIP_VISIBLE_DEVICES"] == "1" and env["ROCR_VISIBLE_DEVICES"] == "2,3"
    env = {}
    pin_local_gpu(env, 5)
    assert env == {"HIP_VISIBLE_DEVICES": "5"}
    with pytest.raises(SystemExit):
        pin_local_gpu({"HIP_VISIBLE_DEVICES": "0,1"}, 3)


def test_tapvid_metrics_of_an_empty_video_are_nan_not_an_exception():
    """eval/metrics.py divides numpy scalars (nan + a warning when a video has no evaluated / visible points)."""
    import math
    from dino_tracker_amd.tapvid import metrics_from_counts
    m = metrics_from_counts([0] * 18)
    assert all(math.isnan(v) for v in m.values())
    m = metrics_from_counts([10, 5, 0] + [0] * 15)
    assert m["occlusion_accuracy"] == 0.5 and math.isnan(m["average_jaccard"])


def test_launcher_refuses_a_path_that_resolves_to_the_reference_before_running_it(tmp_path):
    """the resolution check runs BEFORE the script (it used to run in `finally`, after the whole reference path
    had executed), and an exception of the script itself is not replaced."""
    import subprocess, sys  # noqa: E401
    fake = tmp_path / "models"
    fake.mkdir()
    (tmp_path / "boom.py").write_text("raise ValueError('the script ran')\n")
    r = subprocess.run([sys.executable, "-m", "dino_tracker_amd.run", str(tmp_path / "boom.py")], capture_output=True,
                       text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode != 0 and "the script ran" in r.stderr and "ValueError" in r.stderr
