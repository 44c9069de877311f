#!/bin/bash
# per-kernel rocprofv3 averages of tools/conv_times.py under the dF variants:  gpurun -- 'bash tools/experiments/conv_df_prof.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/conv_df_prof
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for v in "${@:-64}"; do
  IFS=: read -r bkp wpc nst <<< "$v"
  export T4K_CONVBIG_DF8=$bkp; [ -n "${wpc:-}" ] && export T4K_CONVBIG_DF8_WPC=$wpc || unset T4K_CONVBIG_DF8_WPC
  [ -n "${nst:-}" ] && export T4K_CONVBIG_DF8_NST=$nst || unset T4K_CONVBIG_DF8_NST
  python -m pytest "$R/tests/test_gpu_parity.py" -m gpu -q -x -k conv 2>&1 | grep -E "passed|failed"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$O/v" -o c -- python "$R/tools/conv_times.py" > "$O/v.log" 2>&1
  echo "## T4K_CONVBIG_DF8=$bkp WPC=${wpc:-default} NST=${nst:-default}"; tail -4 "$O/v.log" | head -2
  python "$R/tools/rocpd_summary.py" "$(find "$O/v" -name '*.db' | head -1)" | grep -E "convbig_df" | cut -c1-60,112-170
  rm -rf "$O/v"
done
