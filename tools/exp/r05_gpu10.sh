cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
python tools/exp/open_end_error.py 400 24 2>/dev/null > $O/open_end_error.txt
cat $O/open_end_error.txt
