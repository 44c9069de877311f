"""The dataset feed's device prefetch ring (host/dataset.cpp, the reference's TODO src/mu/dataset.cu:112) and the label path that rides in the
conv stack's head forward (one-hot rows + hit flags, Model::onehot(Dataset&) + hit, src/nn/loss.cpp:47-107): a corpus that does NOT divide by the
batch (short last batch: its tail keeps the previous batch's samples, dataset.cu:142-158), rewinds at the end and in the middle of a read-ahead,
a re-normalisation that voids what was staged ahead, more batches than ring buffers.  Forward + nn.hit only (weights fixed), so every number is
a pure function of what the feed delivered: the product VM must print what the oracle VM (CPU, no ring, no rider) prints, with the ring on and off.
Words that ask for host service (dataset, fetch, rewind, normalize, the `next` of a dataset loop) end their input lines: the VM drops what follows, as the reference's does."""
import os
import subprocess

import pytest

from vm_util import ROOT, TEN4, TEN4_ORACLE, compare, run_vm

pytestmark = pytest.mark.gpu

SRC = """0 trace
96 28 28 1 nn.model 0.5 10 conv2d 2 maxpool relu 0.5 20 conv2d 0.5 dropout 2 maxpool relu flatten 100 linear 0.5 dropout 10 linear softmax constant net
96 dataset mnist_train
constant ds0
variable hits 0 hits !
: epoch ( N D -- N ) for forward nn.hit hits +! next ;
: lossy ( N D -- N ) for forward loss.ce . next ;
net ds0 epoch
." e1 " hits @ .
ds0 rewind
drop 0 hits !
ds0 epoch
." e2 " hits @ .
ds0 128 128 normalize
drop 0 hits !
ds0 epoch
." e3 " hits @ .
ds0 rewind
drop ds0 fetch
drop ds0 fetch
drop ds0 fetch
drop ds0 rewind
drop 0 hits !
ds0 epoch
." e4 " hits @ .
ds0 rewind
drop ds0 lossy
bye
"""


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    d = tmp_path_factory.mktemp("feedring")
    subprocess.run(["python3", os.path.join(ROOT, "tools", "make_synth_mnist.py"), os.path.join(str(d), "data", "MNIST", "raw"), "1000", "200"],
                   check=True, capture_output=True)                        # 10 batches of 96 + one of 40: more batches than ring buffers, short tail
    return str(d)


def test_feed_ring_short_tail_rewinds_and_renormalise_match_the_oracle_vm(corpus):
    ref = run_vm(TEN4_ORACLE, source=SRC, seed=11, cwd=corpus)
    assert "e4" in ref and "?" not in ref.replace("-> ok", ""), ref[-800:]
    for env in ({}, {"T4_FEED_PREFETCH": "0"}, {"T4_STACK_HEAD": "0"}):
        got = run_vm(TEN4, source=SRC, seed=11, cwd=corpus, env_extra=env)
        bad = compare(got, ref)
        assert not bad, (env, bad[:5])


TRAIN = """0 trace
96 28 28 1 nn.model 0.5 10 conv2d 2 maxpool relu 0.5 20 conv2d 2 maxpool relu flatten 100 linear 10 linear softmax constant net
96 dataset mnist_train
constant ds0
: epoch ( N D -- N ) for forward backprop 0.01 0.0 nn.sgd next ;
: epochs ( N n -- N ) 1- for ds0 epoch ds0 rewind drop next ;
net 4 epochs
." w0 " 0 nn.w .
." w3 " 3 nn.w sum . drop
." w7 " 7 nn.w sum . drop
." b8 " 8 nn.b .
bye
"""


def test_reader_thread_never_touches_the_deferred_fold(corpus):
    """the conv stack's backward leaves its dF | dB fold to the optimizer launch (T4_OPT_FOLD=1, default) while the feed's reader thread
    waits on events through the C-ABI.  That thread must neither run the fold a second time nor clear the bit: 44 dataset-fed steps with the fold
    inside the optimizer launch print bit for bit what the stand-alone fold (T4_OPT_FOLD=0) prints, repeatedly (the race was intermittent)."""
    want = run_vm(TEN4, source=TRAIN, seed=5, cwd=corpus, env_extra={"T4_OPT_FOLD": "0"})
    assert "w0" in want and "?" not in want.replace("-> ok", ""), want[-600:]
    for rep in range(6):
        got = run_vm(TEN4, source=TRAIN, seed=5, cwd=corpus, env_extra={"T4_OPT_FOLD": "1"})
        assert compare(got, want, rtol=0, atol=0) == [], rep
