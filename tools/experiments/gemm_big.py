import subprocess, sys, os
for v in ("0","1"):
    for shp in ("2048x2048x2048","4096x4096x4096","1024x4096x2048"):
        subprocess.call([sys.executable, "tools/gemm_tune.py", "one", shp], env=dict(os.environ, T4K_GEMM_PLAIN_BIG=v))
