export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=gpurun_out/r6q; mkdir -p $out
rm -rf /tmp/prof_ss; cd /tmp && WH_NO_MAILBOX=1 RUNS=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ss -- python $R/tools/single_stream_prof.py > $R/$out/ss.txt 2>&1
cd $R; f=$(find /tmp/prof_ss -name "*kernel_stats.csv" | head -1); cp $f $out/ss_kernel_stats.csv
grep run_full $out/ss.txt | tail -2; head -40 $out/ss_kernel_stats.csv | cut -c1-200
