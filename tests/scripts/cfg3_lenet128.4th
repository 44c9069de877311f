\ BASELINE config #3 exactly as bench.py runs it: t4_30e `nn_f` LeNet net, batch 128, BOTH dropouts on,
\ `forward backprop 0.01 nn.sgd` x 3 in a compiled loop (the fused launch plan of the product VM), then every
\ parameter tensor, both dropout masks and the loss
0 trace
128 28 28 1 nn.model
0.5 10 conv2d 2 maxpool relu
0.5 20 conv2d 0.5 dropout 2 maxpool relu
flatten 100 linear 0.5 dropout 10 linear softmax
constant net
128 28 28 1 tensor rand constant img
: hot ( T -- T ) 128 0 do 1 i 10 * i 7 * 10 mod + t! loop ;
1280 vector zeros hot 128 1 10 1 reshape4 constant lbl
: fb ( N -- N ) img forward lbl backprop ;
: opt ( N -- N ) 0.01 0.0 nn.sgd ;
: steps ( N n -- N ) 1- for fb opt next ;
net 3 steps
." mask_conv " 4 nn.ex sum . drop
." mask_lin " 9 nn.ex sum . drop
." w0 " 0 nn.w sum . drop ." b0 " 0 nn.b .
." w3 " 3 nn.w sum . drop ." b3 " 3 nn.b sum . drop
." w8 " 8 nn.w sum . drop ." b8 " 8 nn.b sum . drop
." w10 " 10 nn.w sum . drop ." b10 " 10 nn.b .
." w0_all " 0 nn.w .
img forward ." ce " lbl loss.ce . ." hit " nn.hit .
." out " -1 n@ sum . drop
fb ." dw10 " 10 nn.dw .
." db3 " 3 nn.db sum . drop ." dx " 0 n@ sum . drop
drop
bye
