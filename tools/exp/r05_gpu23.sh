for sh in 2 3 4 5; do
  sz=$([ $sh == 4 ] && echo 3 || echo 150)
  for pr in productNiter=2 productNiter=3 productNiter=8 inflateCycles=1 inflateCycles=5 gibbsIters=1 gibbsIters=5 useMsgLikelihoods=1 limitfixeddown=1 "spreadNH=1.0 inflation=2.0" nullSurplusAdd=0.0; do
    python tools/exp/stagewise_any_n.py $sh 200 $sz $pr 2>&1 | grep "^shape" | cut -c1-420
  done
done
