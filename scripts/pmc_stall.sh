# Stall attribution of the MFMA kernels: three rocprofv3 --pmc passes (counters only, no tracing domains) over
# scripts/pmc_stall_driver.py; summarised by scripts/summarise_pmc.py into profiles/<tag>_pmc_stall.json
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/${TAG}_stall_a -o p --output-format csv -- python3 $R/scripts/pmc_stall_driver.py > $R/gpurun_out/${TAG}_stall_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_stall_b -o p --output-format csv -- python3 $R/scripts/pmc_stall_driver.py > $R/gpurun_out/${TAG}_stall_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/${TAG}_stall_c -o p --output-format csv -- python3 $R/scripts/pmc_stall_driver.py > $R/gpurun_out/${TAG}_stall_c.log 2>&1
rm -f $R/gpurun_out/${TAG}_stall_*/*trace.csv $R/gpurun_out/${TAG}_stall_*/*agent*
python3 $R/scripts/summarise_pmc.py $R/gpurun_out ${TAG}_stall_ > $R/gpurun_out/${TAG}_pmc_stall.json
