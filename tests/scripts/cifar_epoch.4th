\ CIFAR-10-shaped corpus (tools/make_synth_cifar.py): the loader's planar RGB -> HWC re-ordering read back pixel by pixel, then one
\ epoch of a small conv + batchnorm net with AdamW and the accuracy on the test split
0 trace
32 dataset cifar10_train
constant ds0
32 dataset cifar10_test
constant ds1
ds0 nn.len ." n_train " . ds1 nn.len ." n_test " . 2drop
32 32 32 3 nn.model 0 16 conv2d 0.01 batchnorm relu 2 maxpool 0 16 conv2d relu 2 maxpool 0.2 dropout flatten 32 linear relu 10 linear softmax constant net
\ pixels (0,0) and (y=16,x=11) of all 32 samples of the first batch (x/256 normalisation), read from the model's input layer after a forward
net ds0 forward
0 n@ 0 1 0 1 slice ." px00 " .
drop 0 n@ 11 12 16 17 slice ." px1611 " .
drop
variable hits 0 hits !
: epoch ( N D -- N ) for forward loss.ce . nn.hit hits +! backprop 0.002 nn.adamw next ;
ds0 epoch
cr
." train_hits " hits @ .
0 hits ! 0 trainable
: test ( N D -- N ) for forward nn.hit hits +! next ;
ds1 test
cr ." test_hits " hits @ .
." w0 " 0 nn.w sum . drop
bye
