for lib in "$@"; do
  echo "== $lib"; SC_HIP_LIB=$PWD/$lib python tools/fused_ab.py 0 2>&1 | grep "C="
done
