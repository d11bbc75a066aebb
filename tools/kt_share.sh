export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for leg in plain p2p rccl; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_$leg -- python $R/tools/dist_overhead.py 108 $leg > $R/gpurun_out/kt_$leg.log 2>&1)
  f=$(find $R/gpurun_out/kt_$leg -name '*kernel_trace.csv' | head -1)
  echo "== $leg"; tail -1 $R/gpurun_out/kt_$leg.log
  python $R/tools/gap_profile.py $f | tee $R/gpurun_out/kt_${leg}_gaps.txt
  # keep only a small slice of the trace
  head -4000 $f > $R/gpurun_out/kt_${leg}_head.csv; rm -rf $R/gpurun_out/kt_$leg
done
