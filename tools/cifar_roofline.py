#!/usr/bin/env python3
"""Roofline line for the CIFAR-shaped step of tools/forth/cifar_steps.4th (the reference's t4_42a-style net: 3 x [conv3x3 + batchnorm + relu +
maxpool + dropout], linear 256 + relu + dropout, linear 10 + softmax; N = 256, AdamW) by SURVEY.md 8(d)'s rule:
    bytes/step = 4 N (4 A + 2 M + S) + 4 P k_opt      FLOP/step = 3 N sum(2 H0 W0 K^2 C1 C0 | 2 E1 E0)
A = layer-boundary activations per image, M = derivative / dropout masks and the batch-norm x-hat, S = inputs re-read by the backward (conv dF,
linear dW, pool arg-max), P = parameters, k_opt = 11 (Adam).      usage: cifar_roofline.py <ms per step>"""
import json, sys
N = 256
convs = [(32, 3, 64), (16, 64, 128), (8, 128, 256)]          # (H = W of the conv grid, C1, C0), 3x3 "same", then 2x2 maxpool
A = 32 * 32 * 3; M = 0; S = 0; P = 0; F = 0
for h, c1, c0 in convs:
    grid, pooled = h * h * c0, (h // 2) ** 2 * c0
    A += 3 * grid + 2 * pooled                 # conv, batchnorm, relu outputs; pool, dropout outputs
    M += grid + grid + pooled                  # batchnorm x-hat, relu mask, dropout mask
    S += h * h * c1 + grid                     # conv input (dF), pool input (arg-max)
    P += 9 * c1 * c0 + c0 + 2 * c0             # filter, bias, gamma / beta
    F += 2 * h * h * 9 * c1 * c0
flat = 4 * 4 * 256
A += flat + 3 * 256 + 2 * 10; M += 2 * 256; S += flat + 256
P += flat * 256 + 256 + 256 * 10 + 10; F += 2 * flat * 256 + 2 * 256 * 10
bytes_step = 4 * N * (4 * A + 2 * M + S) + 4 * P * 11
flop_step = 3 * N * F
ms = float(sys.argv[1]) if len(sys.argv) > 1 else None
out = {"workload": "tools/forth/cifar_steps.4th (t4_42a-style CIFAR net, N=256, AdamW)", "A": A, "M": M, "S": S, "P": P,
       "algorithmic_bytes_per_step": bytes_step, "flop_per_step": flop_step,
       "hbm_floor_ms": round(bytes_step / 8e12 * 1e3, 4), "mfma_floor_ms": round(flop_step / 157.3e12 * 1e3, 4)}
if ms:
    out.update({"ms_per_step": ms, "hbm_frac": round(bytes_step / (ms * 1e-3) / 8e12, 4), "mfma_frac": round(flop_step / (ms * 1e-3) / 157.3e12, 4),
                "bound": "mfma" if flop_step / 157.3e12 > bytes_step / 8e12 else "hbm"})
print(json.dumps(out))
