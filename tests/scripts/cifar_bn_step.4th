\ CIFAR-shaped block stack (conv + batchnorm + relu + maxpool + dropout, as the reference's CIFAR-10 demo builds it),
\ 32x32x3 input, wider channels (MFMA conv path), AdamW steps
0 trace
: cbnr ( M c -- M ) 0 swap conv2d 0.01 batchnorm relu ;
: act ( M p -- M ) >r 2 maxpool r> dropout ;
8 32 32 3 nn.model 16 cbnr 0.25 act 40 cbnr 0.3 act flatten 0 32 linear relu 0 10 linear softmax constant net
net network
8 32 32 3 tensor randn constant img
: hot ( T -- T ) 8 0 do 1 i 10 * i 7 * 3 + 10 mod + t! loop ;
80 vector zeros hot 8 1 10 1 reshape4 constant lbl
: step ( N -- N ) img forward lbl loss.ce . lbl backprop 0.001 nn.adamw ;
: steps ( N n -- N ) 1- for step next ;
4 steps cr
." bn_w " 1 nn.w .
." bn_b " 1 nn.b sum . drop
." c0 " 0 nn.w sum . drop
." c5 " 5 nn.w sum . drop
." l11 " 11 nn.w sum . drop
img forward ." out " -1 n@ .
bye
