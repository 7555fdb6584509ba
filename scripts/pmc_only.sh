R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python3 $R/bench.py --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python3 $R/bench.py --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_write.json 2> $O/pmc_write.err
rm -f $O/*/*kernel_trace.csv $O/*/*agent_info.csv; ls $O/pmc_fetch $O/pmc_write
