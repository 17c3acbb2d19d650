#!/bin/bash
# rocprofv3 kernel trace of the soft-shadow frames (tests/diag/measure_soft_shadows.py, CPU rows skipped) for profiles/.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profile_soft
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export CGRT_NO_CPU=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $R/tests/diag/measure_soft_shadows.py > $out/run.log 2> $out/trace.err || exit 1
cat $out/run.log
cut -c1-200 $out/trace/t_kernel_stats.csv
