#!/bin/bash
# Final GPU jobs of round 6 (1/2): PMC constants per shipped variant + kernel-trace summary of the default workload on the final code.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06
bash tools/profile_r06.sh > gpurun_out/r06/profile.log 2>&1
tail -14 gpurun_out/r06/profile.log
