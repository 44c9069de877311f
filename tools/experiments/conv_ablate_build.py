#!/usr/bin/env python3
"""Ablation builds of the gather-MFMA conv kernel (where do conv2's ~20 us go?): patched copies of csrc/conv.hip under
build/abl/, each linked with the product's other objects into build/abl/libt4hip_<tag>.so.  Timing: conv_ablate_time.py."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = open(os.path.join(ROOT, "tensorforth_amd/csrc/conv.hip")).read()
OUT = os.path.join(ROOT, "build/abl"); os.makedirs(OUT, exist_ok=True)

def rep(s, a, b, count=1):
    assert a in s, a
    return s.replace(a, b, count)

V = {}
V["base"] = SRC
V["nofilt"] = rep(SRC, "            for (int e = tid; e < n4; e += 256) reinterpret_cast<float4 *>(Bl)[e] = reinterpret_cast<const float4 *>(F)[e];",
                  "            for (int e = tid; e < n4; e += 256) reinterpret_cast<float4 *>(Bl)[e] = make_float4(1.f, 1.f, 1.f, 1.f);")
V["noload"] = rep(SRC, "                    const float v = nX[ok ? off[t] + ci : 0];", "                    const float v = (float)(off[t] + ci);")
s = rep(SRC, "                    Y[a] = e;", "")
s = rep(s, "pe->Fpre[a] = f; pe->P[a] = o; e = o;", "e = o;")
s = rep(s, "pe->Fpost[z] = f; pe->R[z] = o; pv = o;", "pv = o;")
s = rep(s, "                if (pe->R2) pe->R2[z] = pv;", "")
V["qonly"] = s                     # epilogue stores only the pooled tensor
V["norng"] = rep(SRC, "        const bool draw = pe->pre == T4K_L_DROPOUT;", "        const bool draw = false;")
V["nomfma"] = rep(SRC, "                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b, acc, 0, 0, 0);", "                        acc[t & 15] += a[u][t] * b;")
objs = [o for o in "runtime elementwise reduce gemm optim linalg fused linear_small comm conv_big".split()]
for tag, src in V.items():
    f = os.path.join(OUT, "conv_%s.hip" % tag); open(f, "w").write(src)
    o = os.path.join(OUT, "conv_%s.o" % tag)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tensorforth_amd/csrc"), "-c", f, "-o", o])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libt4hip_%s.so" % tag), o] +
                          [os.path.join(ROOT, "build/csrc/%s.o" % x) for x in objs] + ["-ldl"])
    print("built", tag, flush=True)
