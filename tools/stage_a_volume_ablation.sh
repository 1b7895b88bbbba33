# Ablation of the anti-phase stage A at 4096 samples against the volume (SC_MTFFT_DEBUG: results wrong, time only): which part of the kernel
# stops scaling at 11 GB of output (profiles/r06_stage_a_volume.txt).  Launch slicing was the other suspect: SC_MTFFT_SLICE=<rounds> python tools/stage_a_volume.py
for d in 0 1 2 3; do echo "== SC_MTFFT_DEBUG=$d (0 whole, 1 no stores, 2 no passes, 3 prologue only)"; SC_MTFFT_DEBUG=$d timeout 300 python tools/stage_a_volume.py 2>&1 | grep -E "N= 4096  1  750|N= 4096  1  250|N= 4096  3   83"; done
