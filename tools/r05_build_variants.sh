#!/bin/bash
# builds the developer variants of the library that round 4 left unmeasured (no GPU needed; ~10 s each):
#   600 SNN_LDS_XTRACE   700 SNN_DIGEST_EARLY   900 SNN_POLL2   950 all three   (500 SNN_DEFER: measured in round 4, 16 % slower)
cd "$(dirname "$0")/.." || exit 1
WHATIF_EXTRA=-DSNN_LDS_XTRACE=1 bash tools/r04_sensitivity_build.sh 600
WHATIF_EXTRA=-DSNN_DIGEST_EARLY=1 bash tools/r04_sensitivity_build.sh 700
WHATIF_EXTRA=-DSNN_POLL2=1 bash tools/r04_sensitivity_build.sh 900
WHATIF_EXTRA="-DSNN_LDS_XTRACE=1 -DSNN_DIGEST_EARLY=1 -DSNN_POLL2=1" bash tools/r04_sensitivity_build.sh 950
rm -f bindsnet_amd/csrc/build/whatif_*.o
ls -la bindsnet_amd/lib/whatif/
