\ LeNet-style CNN (BASELINE configs[2]) on a synthetic 28x28 batch of 128: timed fwd + backprop + SGD steps
0 trace
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
20 steps lbl loss.ce ." warm_ce " .
variable t0
: blk ( N -- N ) clock t0 ! 1000 steps clock t0 @ - ." ms_for_1000 " . cr ;
blk blk blk blk blk blk
bye
