#!/bin/bash
# final check of the GPU suite, then rocprofv3 stats + HBM traffic (separate PMC passes) of the wind kernels and of pv
# family members the bench line does not show -> gpurun_out/summ/r02_extra.{txt,json}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/extra $REPO/gpurun_out/summ
timeout 100 python -m pytest tests -m gpu -q -x --ignore=tests/test_gpu_wind_speed.py > gpurun_out/extra/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/extra/pytest.log
grep -E "passed|failed|rc=" gpurun_out/extra/pytest.log | tail -3
timeout 60 python -m pytest tests/test_gpu_wind_speed.py -m gpu -q > gpurun_out/extra/pytest_wind_speed.log 2>&1
echo "pytest rc=$?" >> gpurun_out/extra/pytest_wind_speed.log
grep -E "passed|failed|rc=|^E  " gpurun_out/extra/pytest_wind_speed.log | tail -8
OUT=$REPO/gpurun_out/prof_r02extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 40 rocprofv3 --kernel-trace --stats -d $OUT/stats -o x -- python $REPO/tools/profile_extra.py > $OUT/stats.log 2>&1
timeout 40 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o x -- python $REPO/tools/profile_extra.py > $OUT/pmc_fetch.log 2>&1
timeout 40 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o x -- python $REPO/tools/profile_extra.py > $OUT/pmc_write.log 2>&1
cd $REPO
python tools/rocpd_summary.py $OUT gpurun_out/summ/r02_extra wind_c3_and_pv_family > /dev/null 2>&1
cp $OUT/stats.log gpurun_out/extra/stats.log
rm -rf $OUT
grep -E "^C3|general|tracking|KANENA" gpurun_out/extra/stats.log | head -12
grep -E "read=" gpurun_out/summ/r02_extra.txt | head -16
