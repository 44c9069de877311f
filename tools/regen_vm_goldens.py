#!/usr/bin/env python3
"""Regenerates tests/golden/vm/*.out (and tests/golden/refhost/*) FROM THE REFERENCE'S OWN HOST CODE.

Build container only.  `make -C oracle refhost` compiles the reference's 17 host sources where they lie under /root/reference/src
and links them with the reference-side binding (integration/t4k_bind*.cpp) over the CPU implementation of include/t4k.h
(oracle/t4k_on_oracle.cpp): oracle/_ref/ten4_refhost is the reference's real VM, printer, layer factory, dataset loaders, model saver
and TensorBoard writer running on the oracle's arithmetic.  Every tests/scripts/*.4th is replayed through it with T4_SEED=1 and its
stdout, normalised as below, becomes the committed golden the product VM is compared with on the GPU (tests/test_vm_scripts.py) and the
oracle VM on the CPU.  Normalisation: the start-up chatter up to the `\\ MMU.stat` line is replaced by the banner line, the `vm0>` trace
lines the reference prints while `0 trace` itself executes are dropped, the tear-down lines become the product's trailer.

Scripts in HAZARD are decided reference hazards (SURVEY.md 9 / DESIGN.md 7): their numbers come from the oracle VM (the product's
host over the oracle) because the reference's code path does not compute anything meaningful there; tests/test_refhost_parity.py still
requires every non-numeric token (layer tables, prompts, shapes) to equal the reference VM's.

Also written: tests/golden/refhost/ref_model_roundtrip.t4 - a model file saved by the reference's aio_model.cpp (f-2 interop fixture:
the product must load it and reproduce the forward output), and tests/golden/refhost/tb_events.tfevents - the tfevents file of the
reference's src/tb writer for tests/scripts_tb/tb_words.4th under a pinned clock (f-4: byte equality)."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFHOST = os.path.join(ROOT, "oracle", "_ref", "ten4_refhost")
FIXTIME = os.path.join(ROOT, "oracle", "_ref", "libfixedtime.so")
ORACLE_VM = os.path.join(ROOT, "oracle", "ten4_oracle")
HAZARD = {"dconv_gen", "hazards"}          # numbers from the oracle VM (see docstring)
TB_TIME = "1700000000"


def normalise_refhost(text):
    i = text.index("\\ MMU.stat")
    body = text[text.index("\n", i) + 1:]
    keep = [l for l in body.splitlines() if not l.startswith("vm0> ")]
    out = []
    for l in keep:
        if l.startswith("\\ VM[] freed"):
            break
        out.append(l)
    return "tensorForth v4.0\n" + "\n".join(out) + "\n\ntensorForth done.\n"


_TRACE_DROP = (r"^vm0> ", r"^NetVM::", r"^} NetVM::", r"^\\ ", r"^tensorForth", r"^VM\[", r"^ForthVM", r"^TensorVM", r"^NetVM", r"^\*\*\* redefined", r"^ *::", r"^\s*$",
               r"^tenvm#", r"^\d+> tenvm#", r"^} tenvm#", r"^} \d+> tenvm#")


def normalise_trace(text):
    """What a run at `1 trace` / `2 trace` prints of the MODEL (src/nn/forward.cu:31-76, backprop.cu:40-107, loss.cpp, model.cpp: input preview, a line per
    layer with its sum per sample and channel, the layer dumps of level 2, the loss derivative, Model::add / loss / onehot / hit lines) - and everything a
    `0 trace` run prints, the optimizer's block included (gradient.cu:19-126: #grad_alloc, per layer and parameter tensor the sums around the update, the
    small tensors' dumps at level 2).  Dropped: the VM-level trace (vm0> stack pushes, NetVM:: / tenvm# word brackets: not on the nn path), the start-up
    chatter.  Masked: the clock fields and the pool offsets #grad_alloc prints."""
    import re
    out = []
    for l in text.split("\n"):
        if any(re.match(p, l) for p in _TRACE_DROP):
            continue
        l = re.sub(r"(w,b\[\d,\d\] )mtum=.*$", r"\1mtum=<pool offsets>", l)              # #grad_alloc: the reference prints the pool offsets of its m / v tensors
        l = re.sub(r"(} Model::(sgd|adam|adamw))\s+-?[\d.]+ ms", r"\1 <t> ms", l)
        l = re.sub(r"^\s*-?\d+\.\d\d:(\s*\d+> )", r"<t>:\1", l)
        l = re.sub(r"(} Model::(forward|backprop))\s+-?[\d.]+ ms", r"\1 <t> ms", l)
        out.append(l.rstrip())
    return "\n".join(out) + "\n"


def normalise_oracle_vm(text):
    lines = text.splitlines()
    return "tensorForth v4.0\n" + "\n".join(lines[1:]) + "\n"


def workdir():
    d = tempfile.mkdtemp(prefix="t4gold")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_mnist.py"), os.path.join(d, "data", "MNIST", "raw"), "1024", "256"], check=True, capture_output=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_cifar.py"), os.path.join(d, "data", "CIFAR10", "cifar-10-batches-bin"), "256", "64"], check=True, capture_output=True)
    for f in glob.glob(os.path.join(ROOT, "tests", "golden", "refhost", "*.t4")):
        shutil.copy(f, d)
    return d


def run(binary, script, cwd, args=(), preload=False, seed=1):
    env = dict(os.environ, T4_SEED=str(seed), T4_TB_FIXED_TIME=TB_TIME)
    if preload:
        env["LD_PRELOAD"] = FIXTIME
    with open(script) as f:
        r = subprocess.run([binary, *args], stdin=f, capture_output=True, text=True, env=env, cwd=cwd, timeout=1800)
    assert r.returncode == 0, (binary, script, r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all", "refhost"], check=True, capture_output=True)
    os.makedirs(os.path.join(ROOT, "tests", "golden", "refhost"), exist_ok=True)
    d = workdir()
    # f-2 fixture first: model_save_load.4th saves model_roundtrip.t4 through the reference's AIO::nsave
    run(REFHOST, os.path.join(ROOT, "tests", "scripts", "model_save_load.4th"), d)
    shutil.copy(os.path.join(d, "model_roundtrip.t4"), os.path.join(ROOT, "tests", "golden", "refhost", "ref_model_roundtrip.t4"))
    shutil.copy(os.path.join(d, "model_roundtrip.t4"), os.path.join(d, "ref_model_roundtrip.t4"))
    for s in sorted(glob.glob(os.path.join(ROOT, "tests", "scripts", "*.4th"))):
        name = os.path.basename(s)[:-4]
        if name in HAZARD:
            out = normalise_oracle_vm(run(ORACLE_VM, s, d))
        else:
            out = normalise_refhost(run(REFHOST, s, d))
        with open(os.path.join(ROOT, "tests", "golden", "vm", name + ".out"), "w") as f:
            f.write(out)
        print("golden", name, "(oracle VM: decided hazard)" if name in HAZARD else "(reference VM)")
    # trace levels: the reference's example t4_30d.4th UNCHANGED (it runs at `2 trace`; README.md:340-371 shows such a log) and cnn_step at `1 trace`
    for tname, src in (("t4_30d", "/root/reference/examples/t4_30d.4th"), ("cnn_step_trace1", os.path.join(ROOT, "tests", "scripts_trace", "cnn_step_trace1.4th"))):
        with open(os.path.join(ROOT, "tests", "golden", "refhost", "trace_" + tname + ".out"), "w") as f:
            f.write(normalise_trace(run(REFHOST, src, d)))
        print("golden trace", tname, "(reference VM)")
    # f-4 fixture: the reference's TensorBoard writer under a pinned clock
    tb = os.path.join(d, "tb")
    os.makedirs(tb)
    run(REFHOST, os.path.join(ROOT, "tests", "scripts_tb", "tb_words.4th"), d, args=["-t" + tb, "-rrun1"], preload=True)
    ev = glob.glob(os.path.join(tb, "run1", "events.out.tfevents.*"))
    assert len(ev) == 1, ev
    shutil.copy(ev[0], os.path.join(ROOT, "tests", "golden", "refhost", "tb_events.tfevents"))
    for extra in sorted(glob.glob(os.path.join(tb, "run1", "*"))):
        if extra != ev[0]:
            with open(extra) as f:
                txt = f.read().replace(tb, "<logdir>")
            with open(os.path.join(ROOT, "tests", "golden", "refhost", "tb_" + os.path.basename(extra)), "w") as f:
                f.write(txt)
    print("fixtures written under tests/golden/refhost/")
    shutil.rmtree(d)


if __name__ == "__main__":
    main()
