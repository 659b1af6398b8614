#!/bin/bash
# tools/sph_ab.sh DIR... : tools/bench_sph.py 128 uniform 2 with the in-tree libraries and each alternative build (SHQ_LIBDIR): the steady-state density
# iteration and the hydro pass.  GPU box, repo root.
for which in tree "$@" tree; do
  if [ "$which" = tree ]; then unset SHQ_LIBDIR; else export SHQ_LIBDIR=$PWD/$which; fi
  python tools/bench_sph.py 128 uniform 2 2>&1 | grep -E "density:|hydro:" | tail -3 | sed "s|^|$which  |"
done
