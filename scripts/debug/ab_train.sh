#!/bin/bash
# same-box A/B of the training-style step: ab/libucnerf_base.so (UCNERF_LIB) against the tree's library, interleaved
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "base: "; UCNERF_LIB=$PWD/ab/libucnerf_base.so python scripts/time_train_step.py
  echo -n "new:  "; python scripts/time_train_step.py
done
echo -n "base zc: "; ZC=1 UCNERF_LIB=$PWD/ab/libucnerf_base.so python scripts/time_train_step.py
echo -n "new zc:  "; ZC=1 python scripts/time_train_step.py
