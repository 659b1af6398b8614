#!/bin/bash
# tools/fft_scratch_offset_probe.sh [offsets...] : does the placement of the transposing pipeline's scratch mesh RELATIVE to the mesh matter?  The
# bench's FFT bracket (last step) for byte offsets of the scratch mesh inside its allocation (SHQ_FFT_SCRATCH_OFFSET), three processes each, the
# default first and last.  GPU box, repo root.
offs=${@:-"-1 0 4096 65536 1052672 33558528 -1"}
for off in $offs; do
  for i in 1 2 3; do
    if [ "$off" = "-1" ]; then unset SHQ_FFT_SCRATCH_OFFSET; else export SHQ_FFT_SCRATCH_OFFSET=$off; fi
    python bench.py --steps 4 --warmup 1 --no-sph --no-cpu-baseline > gpurun_out/so.json 2> gpurun_out/so.err
    python -c "
import json;d=json.load(open('gpurun_out/so.json'));k=d['kernels'];print('offset %9s  step %.2f  fft %.2f  walk %.2f' % ('$off', d['ms_per_step'], k['pm_ms']['fft_pipeline_5_passes_incl_greens_function'], k['tree_walk_ms']), flush=True)"
  done
done
