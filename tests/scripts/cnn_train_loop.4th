\ LeNet-style net with both dropouts: 6 training steps in a compiled loop (exercises graph replay,
\ the device-resident RNG stream and the side-stream gradient work), then Adam on a second model
0 trace
8 28 28 1 nn.model
0.5 10 conv2d 2 maxpool relu
0.5 20 conv2d 0.5 dropout 2 maxpool relu
flatten 100 linear 0.5 dropout 10 linear softmax
constant net
8 28 28 1 tensor rand constant img
: hot ( T -- T ) 8 0 do 1 i 10 * i 3 * 10 mod + t! loop ;
80 vector zeros hot 8 1 10 1 reshape4 constant lbl
: step ( N -- N ) img forward lbl loss.ce . lbl backprop 0.002 0.9 nn.sgd ;
: steps ( N n -- N ) 1- for step next ;
net 6 steps cr
." mask1 " 4 nn.ex sum . drop
." mask2 " 9 nn.ex sum . drop
." w0 " 0 nn.w sum . drop
." w3 " 3 nn.w sum . drop
." w8 " 8 nn.w sum . drop
." w10 " 10 nn.w sum . drop
." b10 " 10 nn.b .
img forward ." out " -1 n@ sum . drop
drop
8 1 16 1 nn.model 12 linear tanh 0.2 dropout 4 linear sigmoid constant mlp
8 1 16 1 tensor randn constant x
8 1 4 1 tensor rand constant y
: astep ( N -- N ) x forward y loss.mse . y backprop 0.01 nn.adam ;
: asteps ( N -- N ) 4 for astep next ;
mlp asteps cr
." aw0 " 0 nn.w sum . drop
." aw3 " 3 nn.w .
bye
