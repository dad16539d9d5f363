mkdir -p gpurun_out/pmc1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o p1 -- python $GRAFT_REPO_ROOT/scripts/pmc_gemm.py 2>&1 | tail -15
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o p2 -- python $GRAFT_REPO_ROOT/scripts/pmc_gemm.py 2>&1 | tail -3
ls -la $GRAFT_REPO_ROOT/gpurun_out/pmc1
