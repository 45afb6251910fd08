cd $GRAFT_REPO_ROOT; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp; timeout 600 python $GRAFT_REPO_ROOT/bench.py 2>&1 | tail -1 | cut -c1-400
