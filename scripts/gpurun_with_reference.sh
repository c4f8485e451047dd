#!/bin/bash
# gpurun with a scratch copy of the reference checkout next to the snapshot (for the tests / benchmarks that run the
# reference's scripts un-modified on the GPU box), removed again as soon as the call returns -- the copy never outlives
# the call and is never committed (.ref_scratch/ is git-ignored).
#   scripts/gpurun_with_reference.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
bash scripts/stage_reference.sh > /dev/null
trap 'bash scripts/stage_reference.sh clean' EXIT
/usr/local/graft/bin/gpurun "$@"
