# tools/coresidency_probe: in one process, heavy instruction classes on a second stream (no second process needed)
R=$GRAFT_REPO_ROOT
cd $R
tools/coresidency_probe ${CELL_S:-1.5}
