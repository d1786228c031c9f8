cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8) > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
bash tools/exp/clique_seam_rate.sh > $O/clique_seam_rate.txt 2>&1
cat $O/clique_seam_rate.txt
