cd /tmp
for v in 0 1 0 1; do
  echo TT_SWAP=$v
  T4K_GEMM_TT_SWAP=$v python $GRAFT_REPO_ROOT/tools/experiments/gemm_layouts.py 1024 1024 1024 2>&1 | tail -1 | cut -c1-200
done
