#!/bin/bash
# what moved between incremental=0 and incremental=1 (same build): SQ counters of the trace launches of one headline frame
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for inc in 0 1; do
  bash tools/pmc_frames.sh r03q_inc$inc scene=obj frames=1 spp=256 incremental=$inc > $O/r03q_pmc_inc$inc.log 2>&1
  cp $O/pmc_r03q_inc$inc/summary.txt $O/r03q_pmc_incremental$inc.txt
done
paste -d'|' <(awk '{print $1,$2,$4}' $O/r03q_pmc_incremental0.txt) <(awk '{print $4}' $O/r03q_pmc_incremental1.txt)
