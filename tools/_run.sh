R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dbcnn --output-format csv -- python3 $R/tools/bench_pcnn.py > $R/gpurun_out/prof_dbcnn.log 2>&1
head -25 $R/gpurun_out/prof_dbcnn/*/*kernel_stats.csv | cut -c1-170
