for lib in build/variants/libsc_scalar_fft.so spectral_connectivity_amd/libsc_hip.so build/variants/libsc_scalar_fft.so spectral_connectivity_amd/libsc_hip.so; do
  echo "== $lib"; SC_HIP_LIB=$PWD/$lib python tools/stage_a_ab.py 0 2>&1 | grep "N="; SC_HIP_LIB=$PWD/$lib SC_AB_LONG=1 python tools/stage_a_ab.py 0 2>&1 | grep "N="
done
