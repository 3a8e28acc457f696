#!/bin/bash
# Run ONE gpurun call with the reference's Python source staged next to the repository (gpurun_ref/, git-ignored), so that the
# tests / bench legs that drive the reference's own environment classes over mjlab_amd (tools/reference_env.py) can run on the
# GPU box, which has no /root/reference.  The copy is removed when the call returns; it is never committed.
# Usage: tools/stage_reference.sh <gpurun timeout seconds> '<command run on the GPU box>'
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
T=${1:?timeout}; shift
rm -rf "$R/gpurun_ref"
mkdir -p "$R/gpurun_ref"
cp -r /root/reference/src "$R/gpurun_ref/src"
find "$R/gpurun_ref" -name "__pycache__" -prune -exec rm -rf {} +
/usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
rc=$?
rm -rf "$R/gpurun_ref"
exit $rc
