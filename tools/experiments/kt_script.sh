# per-kernel table of any Forth script under rocprofv3:  gpurun -- 'bash tools/experiments/kt_script.sh tools/forth/gan_steps.4th [rows]'
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/$1; R=${2:-30}
$GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S | grep -i "ms\|per" | head -5
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S > /tmp/ks.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/ks -name "*.db" | head -1) | head -$R | cut -c1-76,112-150
