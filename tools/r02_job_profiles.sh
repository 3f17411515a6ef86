# rocprofv3 stats + PMC passes for the bench workloads (main, night early-out, star polygons) and the summaries
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/profile_gpu.sh r02main --no-extras > gpurun_out/prof_main.log 2>&1
bash tools/profile_gpu.sh r02night --no-extras --night-skip > gpurun_out/prof_night.log 2>&1
bash tools/profile_gpu.sh r02star --no-extras --shape-kind star > gpurun_out/prof_star.log 2>&1
bash tools/profile_gpu.sh r02starnight --no-extras --shape-kind star --night-skip > gpurun_out/prof_starnight.log 2>&1
cd $REPO
mkdir -p gpurun_out/summ
python tools/rocpd_summary.py gpurun_out/prof_r02main gpurun_out/summ/r02_pv_c2 pv_8760x200x200_100shapes_tessellation > /dev/null
python tools/rocpd_summary.py gpurun_out/prof_r02night gpurun_out/summ/r02_pv_c2_nightskip pv_8760x200x200_100shapes_tessellation_nightskip > /dev/null
python tools/rocpd_summary.py gpurun_out/prof_r02star gpurun_out/summ/r02_pv_c2_star pv_8760x200x200_100shapes_star > /dev/null
python tools/rocpd_summary.py gpurun_out/prof_r02starnight gpurun_out/summ/r02_pv_c2_star_nightskip pv_8760x200x200_100shapes_star_nightskip > /dev/null
rm -rf gpurun_out/prof_r02main gpurun_out/prof_r02night gpurun_out/prof_r02star gpurun_out/prof_r02starnight
ls gpurun_out/summ
