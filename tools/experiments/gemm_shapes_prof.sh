#!/bin/bash
# rocprofv3 kernel-trace averages of the general GEMM path, one shape per run:  gpurun -- 'bash tools/experiments/gemm_shapes_prof.sh r03'
set -u
TAG=${1:-rxx}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/${TAG}_gemm_shapes
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
OUT=$O/summary.txt
echo "# rocprofv3 --kernel-trace --stats, ~0.6 s of launches per shape, averages over the LAST 500 launches (tools/experiments/gemm_one.py M N K tA tB [alpha beta] [bias]);" > "$OUT"
echo "# pct = 2MNK / avg launch time (GEMM kernel + its fold launch, if any) against the 157.3 TFLOP/s fp32 MFMA peak" >> "$OUT"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --stats -d "$O/s$i" -o g -- python "$R/tools/experiments/gemm_one.py" $line > "$O/s$i.log" 2>&1
  DB=$(find "$O/s$i" -name '*.db' | head -1)
  python - "$line" "$DB" >> "$OUT" <<'PY'
import re, sqlite3, sys
f = sys.argv[1].split(); M, N, K = int(f[0]), int(f[1]), int(f[2])
c = sqlite3.connect(sys.argv[2])
names = [r[0] for r in c.execute("select name from kernels where name like '%k_gemm%' or name like '%k_splitk%' group by name having count(*) >= 500 order by sum(duration) desc")]
rows = []
for n in names:      # the last 500 launches of each kernel (clocks ramped, caches warm)
    d = [r[0] for r in c.execute("select duration from kernels where name = ? order by start desc limit 500", (n,))]
    rows.append((n, len(d), sum(d) / len(d), min(d)))
tot = sum(r[2] for r in rows) / 1e3
print("## %-34s  %7.2f us/product  %6.1f TFLOP/s  %5.1f %%" % (sys.argv[1], tot, 2.0 * M * N * K / tot / 1e6, 2.0 * M * N * K / tot / 1e6 / 157.3 * 100))
for n, cnt, avg, mn in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)[:100]
    print("   %-100s calls %5d  avg %7.2f us  min %7.2f us" % (n, cnt, avg / 1e3, mn / 1e3))
PY
  rm -rf "$O/s$i"          # the traces are ~10 MB each; the summary is what is kept
done <<'SHAPES'
1024 1024 1024 0 0
1024 1024 1024 0 1
1024 1024 1024 1 0
1024 1024 1024 1 1
1024 1024 1024 0 1 2.0 -1.0
1024 1024 1024 0 1 bias
1024 1024 784 0 0
1024 1024 784 0 1
1024 1024 784 0 1 bias
1024 1024 1000 0 1
512 1024 1024 0 1
2048 2048 784 0 1
2048 2048 1024 0 1 bias
2048 2048 2048 0 0
4096 4096 1024 0 1 bias
4096 4096 1024 0 0
4096 4096 4096 0 0
SHAPES
cat "$OUT"
