# night-skip diagnostics (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
B="python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity --no-extras"
for c in 16 32 64 128; do
  echo "chunk=$c"; ATLITE_HIP_CHUNK=$c $B --night-skip 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(' night-skip kernel_ms=%.3f min=%.3f' % (j['roofline']['kernel_ms'], j['roofline']['kernel_ms_min']))"
done
bash $REPO/tools/pmc_gpu.sh ns_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-extras --night-skip
bash $REPO/tools/pmc_gpu.sh full_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-extras
bash $REPO/tools/pmc_gpu.sh ns_fetch "FETCH_SIZE" bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-extras --night-skip
bash $REPO/tools/pmc_gpu.sh ns_busy "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_LEVEL_WAVES" bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-extras --night-skip
