#!/bin/bash
# Refresh the rocprofv3 summaries kept under profiles/ (run on the GPU box from the repo root):
#   tools/refresh_profiles.sh r03   ->  gpurun_out/profiles_r03/r03_*.{csv,json}, to be copied into profiles/
# Kernel trace and every counter group are separate runs (--pmc is never combined with a trace domain).
set -e -o pipefail
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py"
S="python3 $ROOT/tools/bench_sph.py 128 uniform 2"
cat_csv() { first=1; for f in "$@"; do if [ $first = 1 ]; then cat "$f"; first=0; else tail -n +2 "$f"; fi; done; }
one() { find "$1" -name "*$2" | head -1; }

rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_stats -o s -- $B --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
cp "$(one $OUT/p_stats kernel_stats.csv)" $OUT/${TAG}_bench256_kernel_stats.csv
python3 $ROOT/tools/pmc_summary.py stats $OUT/${TAG}_bench256_kernel_stats.csv > $OUT/${TAG}_bench256_kernel_stats.json
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p_fetch -o f -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-sph > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p_write -o w -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-sph > $OUT/write.log 2>&1
echo "write pass done"
cat_csv "$(one $OUT/p_fetch counter_collection.csv)" "$(one $OUT/p_write counter_collection.csv)" > $OUT/hbm.csv
python3 $ROOT/tools/pmc_summary.py pmc $OUT/hbm.csv > $OUT/${TAG}_bench256_pmc_hbm.json
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
          --output-format csv -d $OUT/p_sq -o q -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-sph > $OUT/sq.log 2>&1
python3 $ROOT/tools/pmc_summary.py pmc "$(one $OUT/p_sq counter_collection.csv)" > $OUT/${TAG}_bench256_pmc_sq.json
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM \
          --output-format csv -d $OUT/p_sq2 -o q -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-sph > $OUT/sq2.log 2>&1
python3 $ROOT/tools/pmc_summary.py pmc "$(one $OUT/p_sq2 counter_collection.csv)" > $OUT/${TAG}_bench256_pmc_sq2.json
echo "sq passes done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_sphs -o s -- $S > $OUT/sph_stats.log 2>&1
cp "$(one $OUT/p_sphs kernel_stats.csv)" $OUT/${TAG}_sph128_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p_sphf -o f -- $S > $OUT/sph_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p_sphw -o w -- $S > $OUT/sph_write.log 2>&1
cat_csv "$(one $OUT/p_sphf counter_collection.csv)" "$(one $OUT/p_sphw counter_collection.csv)" > $OUT/sph_hbm.csv
python3 $ROOT/tools/pmc_summary.py pmc $OUT/sph_hbm.csv > $OUT/${TAG}_sph128_pmc_hbm.json
echo "sph passes done"
rm -rf $OUT/p_* $OUT/hbm.csv $OUT/sph_hbm.csv
ls -la $OUT
