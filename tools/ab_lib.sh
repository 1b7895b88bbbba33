# A/B of whole libraries on one box: the same stage-A tool once per library, one process each.
# usage: bash tools/ab_lib.sh <lib.so> [<lib.so> ...]      (long windows: SC_AB_LONG=1)
# Variant libraries are built from an earlier commit's source against the current objects, e.g.
#   git show <commit>:spectral_connectivity_amd/csrc/sc_mtfft.hip > /tmp/v.hip   (+ absolute include of sc_common.h)
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -c /tmp/v.hip -o /tmp/v.o
#   hipcc --offload-arch=gfx950 -fPIC -shared $(ls build/obj/*.o | grep -v sc_mtfft.hip) /tmp/v.o -lrocfft -o build/variants/libsc_v.so
for lib in "$@"; do
  echo "== $lib"; SC_HIP_LIB=$PWD/$lib python tools/stage_a_ab.py 0 2>&1 | grep "N="
done
