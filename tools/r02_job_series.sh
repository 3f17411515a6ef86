REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/pmc_gpu.sh ser_f "FETCH_SIZE" tools/bench_configs.py C3 | grep cells
bash tools/pmc_gpu.sh ser_w "WRITE_SIZE" tools/bench_configs.py C3 | grep cells
bash tools/pmc_gpu.sh ser_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" tools/bench_configs.py C3 | grep cells
