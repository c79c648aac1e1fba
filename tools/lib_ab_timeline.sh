#!/bin/bash
# Per-dispatch timeline of one forward under several builds of the library (cache-policy / codegen A/B on whole forwards):
#   bash tools/lib_ab_timeline.sh <out dir> <name>=<library path> ...     ("base" = the shipped library)
O=$1; shift; R=$PWD; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  n=${v%%=*}; l=${v#*=}
  if [ "$n" = base ]; then unset MIMO_HIP_LIB; else export MIMO_HIP_LIB=$R/$l; fi
  (timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$n -o t -- python $R/tools/profile_forward.py > $O/trace_$n.log 2>&1)
  T=$(find $O/trace_$n -name "*kernel_trace.csv" | head -1)
  (cd $R && python tools/trace_forward.py "$T" | head -42 > $O/timeline_$n.txt 2>&1); rm -rf $O/trace_$n
  head -3 $O/timeline_$n.txt
done
