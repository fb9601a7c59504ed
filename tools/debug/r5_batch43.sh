cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
HAD_ROWS=64,130,200,256,768 HAD_SHAPES=vis.fc2,llm.down timeout 300 python3 tools/had_bench.py 2>&1 | grep "M="
