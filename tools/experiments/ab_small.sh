for r in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export CFD_LIB=$PWD/tools/experiments/lib_base.so; else unset CFD_LIB; fi
    for B in 8 32; do python tools/c1_time.py $B 2 2>/dev/null | tail -1 | sed "s/^/$which /"; done
  done
done
