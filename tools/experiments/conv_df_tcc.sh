#!/bin/bash
# L2 (TCC) counters of the many-channel dF kernel:  gpurun -- 'bash tools/experiments/conv_df_tcc.sh 256 32 64 64'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/conv_df_pmc
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# at most four TCC counters per pass (more: "exceeds the capabilities of the hardware", and rocprofv3 then hangs in its signal handler - hence the timeout)
timeout 180 rocprofv3 --kernel-trace -f csv -d "$O/p3" -o c --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- python "$R/tools/experiments/conv_df_one.py" "$@" 40 > "$O/p3.log" 2>&1
python "$R/tools/pmc_summary.py" "$(find "$O/p3" -name '*counter_collection.csv' | head -1)" convbig_df
rm -rf "$O/p3"
