#!/bin/bash
for old in "" 1; do
  echo "== SPX_OLD_WALK=$old"
  env ${old:+SPX_OLD_WALK=1} python tools/sweep.py long 2>&1 | grep -E "auto" | cut -c1-200
  env ${old:+SPX_OLD_WALK=1} python tools/sweep.py longdna 2>&1 | grep -E "\[0/" | cut -c1-200
done
