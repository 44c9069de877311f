"""The deferred dX of a per-layer first layer, LINEAR or CONV (host/model.cpp: bstep i == 0, materialize_dx0, t4k_opt_snapshot).  `in = dX`
(/root/reference/src/nn/backprop.cu:240, :185 for the convolution) leaves dX0 in layer 0; the product computes it when a word asks.  Whatever the script does
between the backward and the read - nothing, an optimizer step (the weights of the BACKWARD must be used), a second backward, a look at the
weight tensor - layer 0 must print what the oracle VM (eager, CPU) prints, and T4_LAZY_DX0=0 must change nothing."""
import numpy as np
import pytest

from vm_util import TEN4, TEN4_ORACLE, run_vm, tokens

pytestmark = pytest.mark.gpu

NET = """0 trace
64 constant N
N 16 16 1 nn.model 256 linear 0.2 leakyrelu 0.3 dropout 64 linear 0.2 leakyrelu 1 linear sigmoid constant D
N 16 16 1 tensor rand constant x
N 1 1 1 tensor ones constant T
: show ( -- ) D 0 n@ dup sum . dup max . min . drop ;
"""
# a convolution the sample-resident stacks do not take (batchnorm follows): the per-layer k_conv* path, whose dX launch is the one deferred
NET_CONV = """0 trace
32 constant N
N 12 12 3 nn.model 0.5 8 conv2d batchnorm relu 2 maxpool flatten 10 linear softmax constant D
N 12 12 3 tensor rand constant x
N 10 1 1 tensor rand constant T
: show ( -- ) D 0 n@ dup sum . dup max . min . drop ;
"""
# the same with a full MFMA tile of output channels: forward / dF | dB on the thin-input kernels (csrc/conv_img.hip), layer-0 copy from the conv launch
NET_CONV64 = NET_CONV.replace("0.5 8 conv2d", "0.5 64 conv2d")
NETS = {"linear": NET, "conv": NET_CONV, "conv64": NET_CONV64}
CASES = {
    "read-after-backprop": "D x forward T backprop show",
    "read-after-adam": "D x forward T backprop 0.001 0.5 nn.adam show",
    "read-after-sgd": "D x forward T backprop 0.01 0.0 nn.sgd show",
    "two-backprops-then-adam": "D x forward T backprop x forward T backprop 0.001 0.5 nn.adam show",
    "second-step-keeps-first-offer-closed": "D x forward T backprop 0.001 0.5 nn.adam x forward T backprop 0.001 0.5 nn.adam show",
    "weights-inspected-first": "D x forward T backprop D 0 nn.w sum . drop 0.001 0.5 nn.adam show",
    "frozen": "D 0 trainable x forward T backprop show",
}


def _nums(txt):
    out = []
    for t in tokens(txt):
        try:
            out.append(float(t))
        except ValueError:
            pass
    return np.array(out)


@pytest.mark.parametrize("net", list(NETS))
@pytest.mark.parametrize("name", list(CASES))
def test_layer0_gradient_matches_the_oracle_vm_whenever_it_is_read(name, net):
    src = NETS[net] + CASES[name] + "\nbye\n"
    ref = _nums(run_vm(TEN4_ORACLE, source=src, seed=7))
    for env in ({}, {"T4_LAZY_DX0": "0"}):
        got = _nums(run_vm(TEN4, source=src, seed=7, env_extra=env))
        assert got.shape == ref.shape and got.size >= 3, (name, env, got, ref)
        scale = np.maximum(np.abs(ref), 1e-3)
        assert np.all(np.abs(got - ref) <= 2e-3 * scale + 1e-4), (name, env, got, ref)     # printed at 4 decimals / 6 significant digits


def test_a_deferred_conv_dx_saves_its_launch_and_a_read_brings_it_back(t4k):
    """the backward of the conv-first net leaves the k_conv_dx* launch out;
    reading layer 0 afterwards costs exactly that launch"""
    import ctypes

    from tensorforth_amd.vm import VM
    t4k.lib.t4k_launch_count.restype = ctypes.c_ulonglong
    vm = VM(device=0, seed=7)
    try:
        vm.eval(NET_CONV.replace("constant D", "constant D2").replace("D 0 n@", "D2 0 n@"))
        vm.eval(": st D2 x forward T backprop 0.001 0.5 nn.adam drop ;")
        vm.eval("st st"); t4k.call("t4k_sync", None)
        l0 = t4k.lib.t4k_launch_count(); vm.eval("st"); lazy = t4k.lib.t4k_launch_count() - l0
        vm.eval("D2 x forward T backprop drop")
        l0 = t4k.lib.t4k_launch_count(); vm.eval("D2 0 n@ sum drop drop drop"); first = t4k.lib.t4k_launch_count() - l0
        l0 = t4k.lib.t4k_launch_count(); vm.eval("D2 0 n@ sum drop drop drop"); again = t4k.lib.t4k_launch_count() - l0
        assert first - again in (1, 2), (lazy, first, again)                    # the deferred dX (+ its copy over x), once, when a word looks at layer 0
    finally:
        vm.close()
