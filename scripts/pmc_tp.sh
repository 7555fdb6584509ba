R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/pmctp_a -o p --output-format csv -- python3 $R/scripts/bench_conv.py deconv8s4 3 2 fwd 4 > $R/gpurun_out/pmctp_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC -d $R/gpurun_out/pmctp_b -o p --output-format csv -- python3 $R/scripts/bench_conv.py deconv8s4 3 2 fwd 4 > $R/gpurun_out/pmctp_b.log 2>&1
rm -f $R/gpurun_out/pmctp_*/*trace.csv $R/gpurun_out/pmctp_*/*agent*
