cd "$GRAFT_REPO_ROOT"; N=4 B=64 bash tools/r05_repeat.sh cur fuse fuse2 2>&1 | tail -3; N=3 B=1 bash tools/r05_repeat.sh cur fuse fuse2 2>&1 | tail -3
