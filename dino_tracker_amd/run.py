"""Launcher that runs one of the reference's scripts UNCHANGED on top of this implementation:

    python -m dino_tracker_amd.run [--path DIR]... /path/to/dino-tracker/inference_grid.py --config ... --data-path ...

Why a launcher: `python script.py` puts the script's own directory at sys.path[0], AHEAD of $PYTHONPATH.  The
reference's `models/`, `data/` are namespace packages without `__init__.py`, and for a script that sits in the
reference's root (inference_grid.py, inference_benchmark.py, train.py) that first entry makes `models/tracker.py`,
`models/model_inference.py`, `data/dataset.py` and `utils.py` resolve to the REFERENCE files: the run would silently
execute the PyTorch code.  Here the import path is set explicitly -- overlay first -- before the script starts:

    sys.path = [<repo>/overlay, <repo>, --path entries..., <script dir>, rest]

and after the script has been loaded the launcher checks that the hot-path modules it imported came from the overlay
(a run that does not use this implementation fails loudly instead of succeeding slowly).
"""
from __future__ import annotations

import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OVERLAY = os.path.join(REPO, "overlay")
HOT_MODULES = ("models.tracker", "models.model_inference", "models.extractor", "models.networks.tracker_head",
               "models.networks.delta_dino", "models.networks.conv_norm", "data.dataset", "utils")


def _usage() -> "NoReturn":
    raise SystemExit("usage: python -m dino_tracker_amd.run [--path DIR]... script.py [script arguments]")


def configure_path(script: str, extra: list) -> None:
    script_dir = os.path.dirname(os.path.abspath(script))
    head = [OVERLAY, REPO] + [os.path.abspath(p) for p in extra] + [script_dir]
    # the reference's scripts live either in its root or one level below (preprocessing/*.py import the root's
    # modules through PYTHONPATH in the reference's own instructions): add the root too when it looks like one
    parent = os.path.dirname(script_dir)
    if not os.path.isdir(os.path.join(script_dir, "models")) and os.path.isdir(os.path.join(parent, "models")):
        head.append(parent)
    rest = [p for p in sys.path if p and os.path.abspath(p) not in head]
    sys.path[:] = head + rest


def check_resolution() -> None:
    """Every hot-path module that has been imported must come from the overlay."""
    wrong = []
    for name in HOT_MODULES:
        mod = sys.modules.get(name)
        path = getattr(mod, "__file__", None) if mod is not None else None
        if path and not os.path.abspath(path).startswith(OVERLAY + os.sep):
            wrong.append(f"{name} -> {path}")
    if wrong:
        raise SystemExit("dino_tracker_amd.run: these modules did not resolve to the overlay:\n  " + "\n  ".join(wrong))


def precheck_resolution() -> None:
    """Where WOULD the hot-path modules come from with the path as configured?  (importlib.util.find_spec imports the
    parent packages -- PEP 420 namespaces here -- but not the modules themselves.)"""
    import importlib.util
    wrong = []
    for name in HOT_MODULES:
        try:
            spec = importlib.util.find_spec(name)
        except (ImportError, ValueError):
            spec = None
        origin = getattr(spec, "origin", None) if spec is not None else None
        if origin and not os.path.abspath(origin).startswith(OVERLAY + os.sep):
            wrong.append(f"{name} -> {origin}")
    if wrong:
        raise SystemExit("dino_tracker_amd.run: these modules would not resolve to the overlay:\n  " + "\n  ".join(wrong))


def main(argv: list) -> None:
    extra = []
    while argv and argv[0] == "--path":
        if len(argv) < 2:
            _usage()
        extra.append(argv[1])
        argv = argv[2:]
    if not argv or argv[0].startswith("-"):
        _usage()
    script = argv[0]
    if not os.path.isfile(script):
        raise SystemExit(f"dino_tracker_amd.run: no such script: {script}")
    configure_path(script, extra)
    for name in list(sys.modules):  # nothing of the reference's module names may be cached from before
        if name in HOT_MODULES or name in ("models", "data", "models.networks"):
            del sys.modules[name]
    sys.argv = [script] + argv[1:]
    precheck_resolution()  # before the script runs: a mis-resolved path must not execute the PyTorch reference first
    ok = False
    try:
        runpy.run_path(script, run_name="__main__")
        ok = True
    finally:
        if ok:
            check_resolution()
        else:  # never replace the script's own exception / traceback
            try:
                check_resolution()
            except SystemExit as e:
                print(e, file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1:])
