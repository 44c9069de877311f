\ CIFAR-shaped stack with 32 / 64 channels: the second conv layer takes the LDS-staged many-channel kernels (forward, dX, dF)
0 trace
: cbnr ( M c -- M ) 0 swap conv2d 0.01 batchnorm relu ;
: act ( M p -- M ) >r 2 maxpool r> dropout ;
4 16 16 3 nn.model 32 cbnr 0.2 act 64 cbnr 0.3 act flatten 0 24 linear relu 0 10 linear softmax constant net
4 16 16 3 tensor randn constant img
40 vector zeros 1 2 t! 1 17 t! 1 23 t! 1 39 t! 4 1 10 1 reshape4 constant lbl
: step ( N -- N ) img forward lbl loss.ce . lbl backprop 0.001 nn.adam ;
: steps ( N n -- N ) 1- for step next ;
net 3 steps cr
." c0 " 0 nn.w sum . drop
." c5 " 5 nn.w sum . drop
." bn6 " 6 nn.w sum . drop
." l11 " 11 nn.w sum . drop
img forward ." out " -1 n@ .
bye
