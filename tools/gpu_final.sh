cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/pmc_all.sh r06q cfg5l 2>&1 | tail -12
bash tools/gpu_full.sh
