#!/bin/bash
# builds bindsnet_amd/lib/whatif/libsnnhip_w<k>.so for k in "$@": the product library with -DSNN_WHATIF=k in snn_dc2015_async.hip
# (a delay of 512 clocks at point k of the compute loop; see WHATIF_DELAY there).  Developer aid; the directory is git-ignored.
cd "$(dirname "$0")/../bindsnet_amd/csrc" || exit 1
mkdir -p ../lib/whatif build
OBJS=$(ls build/*.o | grep -v snn_dc2015_async | grep -v whatif | tr '\n' ' ')
for k in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -DSNN_WHATIF=$k $WHATIF_EXTRA -c snn_dc2015_async.hip -o build/whatif_$k.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS build/whatif_$k.o -ldl -o ../lib/whatif/libsnnhip_w$k.so && echo built $k ) &
done
wait
