set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/profile_r05.sh > gpurun_out/r05/profile.log 2>&1
tail -14 gpurun_out/r05/profile.log
OCCS="1 2" bash tools/pmc_occ2.sh > gpurun_out/r05/pmc_occ2.log 2>&1; tail -3 gpurun_out/r05/pmc_occ2.log
