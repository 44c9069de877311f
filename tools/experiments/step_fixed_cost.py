#!/usr/bin/env python3
"""What a timed region of K LeNet steps costs beyond K x (sustained step): host clock around `K steps` + device sync, K = 0 ... 200 (the driver times 20 steps behind 5 warm-ups)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensorforth_amd.vm import VM
vm = VM(device=0, seed=1)
pre = """0 trace
128 28 28 1 nn.model
0.5 10 conv2d 2 maxpool relu
0.5 20 conv2d 0.5 dropout 2 maxpool relu
flatten 100 linear 0.5 dropout 10 linear softmax
constant net
128 28 28 1 tensor rand constant img
: hot ( T -- T ) 128 0 do 1 i 10 * i 10 mod + t! loop ;
1280 vector zeros hot 128 1 10 1 reshape4 constant lbl
: step ( N -- N ) img forward lbl backprop 0.01 0.0 nn.sgd ;
: steps ( N n -- N ) 1- for step next ;
net
50 steps
"""
vm.eval(pre); torch.cuda.synchronize()
for K in (1, 5, 20, 50, 200, 1000):
    best = 1e9
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); vm.eval("%d steps\n" % K); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("K=%4d: %8.1f us total, %.2f us/step" % (K, best * 1e6, best * 1e6 / K), flush=True)
# the driver's shape: an idle device, 5 warm-up steps, ONE timed region of 20 - against the same behind ~0.2 s of dense products (clocks up)
def region(K=20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); vm.eval("%d steps\n" % K); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6 / K
vm.eval("1024 1024 matrix rand constant ma 1024 1024 matrix rand constant mb : mx 1- for matmul drop next ;\n")
for rep in range(3):
    time.sleep(0.5); vm.eval("5 steps\n"); a = region()
    time.sleep(0.5); vm.eval("ma mb 10000 mx 2drop\n"); vm.eval("5 steps\n"); b = region()
    print("idle -> 5 warm-ups -> 20 timed: %.2f us/step     0.2 s of 1024^3 products -> 5 warm-ups -> 20 timed: %.2f us/step" % (a, b), flush=True)
