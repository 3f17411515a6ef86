#!/bin/bash
# rocprofv3 stats + HBM traffic (separate PMC passes) of the per-cell pv kernels -> gpurun_out/summ/r02_pv_c2_cells.{txt,json}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r02cells
mkdir -p $OUT $REPO/gpurun_out/summ
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o pv -- python $REPO/tools/profile_cells.py > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pv -- python $REPO/tools/profile_cells.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pv -- python $REPO/tools/profile_cells.py > $OUT/pmc_write.log 2>&1
cd $REPO
python tools/rocpd_summary.py $OUT gpurun_out/summ/r02_pv_c2_cells pv_8760x200x200_per_cell_kernels
rm -rf $OUT
cat gpurun_out/summ/r02_pv_c2_cells.txt | head -60
