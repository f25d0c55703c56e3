# round 6, call 8: instruction counters of the 61-state walk: k_walkb, k_walkg on bf16 tables (MBAMD_NO_WALKB=1), round 5's k_walkg
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c8; export TMPDIR=/tmp
export PMC_PASSES="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU;SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS;SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU;GRBM_GUI_ACTIVE GRBM_COUNT"
PMC_TAG=_walkb bash tools/pmc_walk.sh c5 > /dev/null 2>&1
MBAMD_NO_WALKB=1 PMC_TAG=_walkg_bf bash tools/pmc_walk.sh c5 > /dev/null 2>&1
MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_r5.so PMC_TAG=_r5 bash tools/pmc_walk.sh c5 > /dev/null 2>&1
for t in _walkb _walkg_bf _r5; do echo "#### $t"; grep -A1 '== PMC\|k_walk' gpurun_out/pmc_walk_c5$t.log | grep 'PMC\|k_walk'; done | tee gpurun_out/c8/pmc.txt
