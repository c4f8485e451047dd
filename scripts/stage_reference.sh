#!/bin/bash
# Put a SCRATCH copy of the reference checkout where a gpurun call can see it (gpurun snapshots /root/repo; the
# GPU box has no /root/reference).  The copy lives in .ref_scratch/ (git-ignored, never committed) and is removed with
# `scripts/stage_reference.sh clean`.  On the box:  DTK_REFERENCE_ROOT=$GRAFT_REPO_ROOT/.ref_scratch/reference
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "clean" ]; then rm -rf .ref_scratch; exit 0; fi
SRC=${DTK_REFERENCE_ROOT:-/root/reference}
mkdir -p .ref_scratch
rm -rf .ref_scratch/reference
cp -r "$SRC" .ref_scratch/reference
rm -rf .ref_scratch/reference/.git
du -sh .ref_scratch/reference
