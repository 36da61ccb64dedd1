# Build an A/B variant of libccgp into build_ab/libccgp_<name>.so with extra compiler flags, e.g.
#   bash scripts/build_variant.sh tab -DCCGP_SMALL_EXP_TABLE=1
#   bash scripts/build_variant.sh occ3 -DCCGP_SMALL_OCC_G8=3
# and compare on ONE box:  gpurun -- 'bash scripts/r03g.sh <tag> <name> [<name> ...]'   (CCGP_LIB selects the library)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/convex-combination-of-gaussian-processes_amd/csrc
OUT=$ROOT/build_ab
mkdir -p $OUT/obj_$NAME
for f in capi cov small small_reg blocked; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result "$@" -c $CSRC/$f.hip -o $OUT/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -c $CSRC/special.cpp -o $OUT/obj_$NAME/special.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -pthread -c $CSRC/multi.cpp -o $OUT/obj_$NAME/multi.o
(cd $OUT/obj_$NAME && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 capi.o cov.o small.o small_reg.o blocked.o special.o multi.o -pthread -o $OUT/libccgp_$NAME.so)
rm -rf $OUT/obj_$NAME
ls -la $OUT/libccgp_$NAME.so
