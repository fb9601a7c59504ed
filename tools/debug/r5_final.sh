cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
bash tools/r5_profiles.sh ac74469 2>&1 | tail -8
