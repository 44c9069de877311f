\ PCIe-inclusive rate: LeNet-style CNN trained from the dataset words (IDX file -> host -> HBM u8 -> on-GPU normalise),
\ one epoch of mini-batches of 128 incl. the per-batch hit count read-back; needs ./data/MNIST/raw (tools/make_synth_mnist.py)
0 trace
128 28 28 1 nn.model
0.5 10 conv2d 2 maxpool relu
0.5 20 conv2d 0.5 dropout 2 maxpool relu
flatten 100 linear 0.5 dropout 10 linear softmax
constant net
128 dataset mnist_train
constant ds0
variable hits 0 hits !
: epoch ( N D -- N ) for forward nn.hit hits +! backprop 0.01 0.0 nn.sgd next ;
: epochs ( N n -- N ) 1- for ds0 epoch ds0 rewind drop next ;   \ a dataset loop asks for host service: the rest of an INTERPRETED line is dropped (as in the reference), so the epochs live in a word
net 1 epochs        \ warm-up epoch
variable t0 clock t0 !
0 hits !
3 epochs
clock t0 @ - ." ms_for_3_epochs " . ." hits " hits @ .
bye
