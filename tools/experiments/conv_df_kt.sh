#!/bin/bash
# kernel-trace time of the dF partial kernel alone at several batch sizes:  gpurun -- 'bash tools/experiments/conv_df_kt.sh H C1 C0 N1 N2 ...'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
H=$1; C1=$2; C0=$3; shift 3
for N in "$@"; do
  rm -rf /tmp/dfkt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dfkt -o c -- python "$R/tools/experiments/conv_df_one.py" $N $H $C1 $C0 300 > /tmp/dfkt.log 2>&1
  echo "N=$N"; python "$R/tools/rocpd_summary.py" "$(find /tmp/dfkt -name '*.db' | head -1)" | grep -E "convbig|fold|colsum" | cut -c1-50,112-150
done
