cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_r03_profiles.sh 2>&1 | tail -20
