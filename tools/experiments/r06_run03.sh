#!/bin/bash
# round 6: timeline of the per-substep schedule on the 65-plane self ring with the transposed interior (rocprofv3 kernel trace)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for xp in 1 0; do
  rm -rf /tmp/prof_$xp
  HJ_XP=$xp rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$xp -- python3 $R/tools/thin_slab_ring.py 513 8 sub > $R/gpurun_out/r06_tl_$xp.log 2>&1 || { tail -5 $R/gpurun_out/r06_tl_$xp.log; exit 1; }
  echo "== HJ_XP=$xp" >> $R/gpurun_out/r06_timeline.txt
  grep "^N=" $R/gpurun_out/r06_tl_$xp.log >> $R/gpurun_out/r06_timeline.txt
  python3 $R/tools/timeline.py /tmp/prof_$xp 30 30 >> $R/gpurun_out/r06_timeline.txt
done
cat $R/gpurun_out/r06_timeline.txt
