for env in "X=1" "SHQ_MAIN_PRIO_DEFAULT=1" "SHQ_FFT_GRID_MUL=32" "SHQ_FFT_GRID_MUL=2"; do
env $env SHQ_DEBUG_PRIO=1 python bench.py --steps 3 --warmup 1 --no-sph --no-cpu-baseline > gpurun_out/b3.json 2> gpurun_out/b3.err; grep -m1 "stream priorities" gpurun_out/b3.err; python -c "
import json;d=json.load(open(\"gpurun_out/b3.json\"));k=d[\"kernels\"];print(\"$env\", round(d[\"ms_per_step\"],2),round(k[\"tree_walk_ms\"],2),round(k[\"resident_full_step_ms\"],2),k[\"pm_ms\"][\"total\"]);print(k[\"resident_timeline_ms\"])"; done
