#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
N=${1:-30}
for L in "" "_nopl"; do
  export EPN_LIB=$R/epn_pointcloud_amd/libepn_so3conv$L.so
  echo "lib '$L': $(bash tools/nan_hunt4.sh $N | grep -c "False") of $N runs non-finite"
done
