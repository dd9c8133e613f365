cd $GRAFT_REPO_ROOT
for r in 16 8 4 1; do echo "== DMX_BIG_ISLAND_ROWS=$r"; DMX_BIG_ISLAND_ROWS=$r bash scripts/time_compat.sh 2>&1 | tail -3; DMX_BIG_ISLAND_ROWS=$r bash scripts/time_compat.sh 2>&1 | tail -3; done
