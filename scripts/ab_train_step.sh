#!/bin/bash
# Same-box A/B of the training-style step: the tree in build/ab_old (an older commit, built there) against this tree, interleaved.
#   RAYS / SAMPLES as scripts/time_train_step.py; also the rendering() training step (scripts/time_dropin.py, MODE=train)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2 3; do
  echo "old: $(python $R/build/ab_old/scripts/time_train_step.py | tail -1)"
  echo "new: $(python $R/scripts/time_train_step.py | tail -1)"
done
for i in 1 2; do
  echo "old dropin: $(MODE=train python $R/build/ab_old/scripts/time_dropin.py | tail -1)"
  echo "new dropin: $(MODE=train python $R/scripts/time_dropin.py | tail -1)"
done
