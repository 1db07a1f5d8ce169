#!/bin/bash
# Same-box A/B of the step with the small launches folded into the coarse pass (FOLD=1, round 4) against the old launch structure (FOLD=0), interleaved:
#   scripts/ab_fold.sh [rays...]
R=$GRAFT_REPO_ROOT
for n in "${@:-512}"; do
  for rep in 1 2; do
    for rp in ${REPACKS:-0}; do
      for f in ${FOLDS:-0 1 2 3}; do RAYS=$n REPACK=$rp FOLD=$f STEPS=${STEPS:-400} python3 $R/scripts/time_strong.py 2>/dev/null || exit 1; done
    done
  done
done
