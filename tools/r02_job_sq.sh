# SQ counter passes for the C2 bench (night-skip and full); usage: bash tools/r02_job_sq.sh [extra bench args]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
C2="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_SALU"
for ns in "--night-skip" ""; do
  bash $REPO/tools/pmc_gpu.sh sq1$ns "$C1" bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-extras $ns "$@" | grep fused
  bash $REPO/tools/pmc_gpu.sh sq2$ns "$C2" bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-extras $ns "$@" | grep fused
done
