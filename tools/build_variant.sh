#!/bin/bash
# An A/B variant of the library: tools/build_variant.sh NAME "-DFLAG=.. -DFLAG=.."  ->  variants/NAME/libfirework_hip.so
# (built from a copy of the sources, so the in-tree build is untouched; tools load it through FW_LIB_PATH)
set -e
NAME=$1; EXTRA=$2
R=$(cd "$(dirname "$0")/.." && pwd); T=/tmp/fw_variant_$NAME
rm -rf $T; mkdir -p $T/bevy_firework_amd $T/include
cp -r $R/bevy_firework_amd/csrc $T/bevy_firework_amd/; cp $R/include/*.h $T/include/
rm -rf $T/bevy_firework_amd/csrc/build $T/bevy_firework_amd/csrc/*.so
make -C $T/bevy_firework_amd/csrc -j6 libfirework_hip.so EXTRA="$EXTRA" > $T/build.log 2>&1
mkdir -p $R/variants/$NAME; cp $T/bevy_firework_amd/csrc/libfirework_hip.so $R/variants/$NAME/
echo "variants/$NAME/libfirework_hip.so  ($EXTRA)"
