for B in 1 2 4 6; do
  python tools/c1_time.py $B 2  | tail -1 | sed "s/^/nfb1 /"
  CFD_RT_NFB2_TILES=1 python tools/c1_time.py $B 2  | tail -1 | sed "s/^/nfb2 /"
done
