#!/bin/bash
# Builds ablated copies of the library on the GPU box and times the forward layers (results are garbage by design).
set -e
cd "$(dirname "$0")/.."
S=epn_pointcloud_amd/csrc
for V in ${ABL:-1 2}; do
  O=gpurun_out/abl$V; mkdir -p $O
  for f in index_kernels conv_generic c_api intra_mfma inter_c1; do cp epn_pointcloud_amd/build/$f.o $O/ 2>/dev/null || hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $S/$f.hip -o $O/$f.o; done
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DEPN_ABLATE=$V -c $S/inter_mfma.hip -o $O/inter_mfma.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib.so $O/*.o
  echo "== ablation $V"; EPN_LIB=$PWD/$O/lib.so python tools/bench_layers.py --iters 5 --only fwd 2>&1 | grep -E "^L[15] |total"
done
echo "== baseline"; python tools/bench_layers.py --iters 5 --only fwd 2>&1 | grep -E "^L[15] |total"
