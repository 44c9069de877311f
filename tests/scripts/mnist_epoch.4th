\ dataset-driven training: CNN on the synthetic MNIST-shaped corpus (tools/make_synth_mnist.py), one epoch of
\ forward / loss / hit / backprop / Adam per mini-batch, then accuracy on the test split
0 trace
64 28 28 1 nn.model 0.5 8 conv2d 2 maxpool relu flatten 32 linear relu 10 linear softmax constant net
64 dataset mnist_train
constant ds0
64 dataset mnist_test
constant ds1
variable hits 0 hits !
: epoch ( N D -- N ) for forward loss.ce . nn.hit hits +! backprop 0.002 nn.adam next ;
net ds0 epoch
cr
." train_hits " hits @ .
ds0 rewind
drop
0 hits ! 0 trainable
: test ( N D -- N ) for forward nn.hit hits +! next ;
ds1 test
cr ." test_hits " hits @ .
." w0 " 0 nn.w sum . drop
bye
