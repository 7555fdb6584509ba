R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA -d $R/gpurun_out/pmcw_a -o p --output-format csv -- python3 $R/scripts/bench_conv.py conv8s4 3 2 wgrad 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM -d $R/gpurun_out/pmcw_b -o p --output-format csv -- python3 $R/scripts/bench_conv.py conv8s4 3 2 wgrad 4 > /dev/null 2>&1
rm -f $R/gpurun_out/pmcw_*/*trace.csv $R/gpurun_out/pmcw_*/*agent*
