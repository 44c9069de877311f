\ CIFAR-10-shaped corpus (tools/make_synth_cifar.py): the loader's planar RGB -> HWC re-ordering read back pixel by pixel, then one
\ epoch of a small conv + batchnorm net with AdamW and the accuracy on the test split
0 trace
32 dataset cifar10_train constant ds0
32 dataset cifar10_test constant ds1
ds0 nn.len ." n_train " . ds1 nn.len ." n_test " . 2drop
\ pixels of sample 0 and 5 of the first batch (x/256 normalisation): (y,x,c) flat index = (y*32 + x)*3 + c
ds0 ." px " 0 t@ . 1 t@ . 2 t@ . 1571 t@ . 3071 t@ . 15360 t@ . 16931 t@ . drop
32 32 32 3 nn.model 0 16 conv2d 0.01 batchnorm relu 2 maxpool 0 16 conv2d relu 2 maxpool 0.2 dropout flatten 32 linear relu 10 linear softmax constant net
variable hits 0 hits !
: epoch ( N D -- N ) for forward loss.ce . nn.hit hits +! backprop 0.002 nn.adamw next ;
net ds0 epoch cr
." train_hits " hits @ .
0 hits ! 0 trainable
: test ( N D -- N ) for forward nn.hit hits +! next ;
ds1 test cr ." test_hits " hits @ .
." w0 " 0 nn.w sum . drop
bye
