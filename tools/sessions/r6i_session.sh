#!/bin/bash
# Round 6, session I: the records -- cpu_baseline thread sweep on the box's host, PMC traffic at the TIMED batch size (448 windows), kernel trace of the bench.
out=gpurun_out/${1:-r6i}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
date
timeout 900 python tools/cpu_sweep.py medium 8 16 32 64 2>&1 | tee $out/cpu_sweep.txt | tail -6
date
cd /tmp && PMC_WINDOWS=448 PMC_ALGO=$R/$out/pmc_algo.json timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python $R/tools/pmc_probe.py > $R/$out/pmc_fetch.log 2>&1
PMC_WINDOWS=448 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python $R/tools/pmc_probe.py > $R/$out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $out/pmc_algo.json $out/r_pmc.json 2>&1 | tail -12
date
rm -rf /tmp/prof_bench
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --inflight 1 --steps 5 --warmup 1 --no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-workloads --no-small-job --no-ids-check > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
cd $R
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats.csv 2>/dev/null
head -14 $out/bench_kernel_stats.csv | cut -c1-150
date
