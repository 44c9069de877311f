\ CIFAR-10-shaped CNN in the style of the reference's t4_42a demo (3 x [conv + batchnorm + relu + maxpool + dropout], linear head),
\ N = 256, 32x32x3 synthetic HBM-resident batch, AdamW; timed training steps
0 trace
256 constant N
: cbnr ( M c -- M ) 0 swap conv2d 0.01 batchnorm relu ;
: act ( M p -- M ) >r 2 maxpool r> dropout ;
N 32 32 3 nn.model 64 cbnr 0.25 act 128 cbnr 0.30 act 256 cbnr 0.40 act flatten 0 256 linear relu 0.5 dropout 0 10 linear softmax constant net
N 32 32 3 tensor randn constant img
: hot ( T -- T ) N 0 do 1 i 10 * i 7 * 3 + 10 mod + t! loop ;
N 10 * vector zeros hot N 1 10 1 reshape4 constant lbl
: step ( N -- N ) img forward lbl backprop 0.001 nn.adamw ;
: steps ( N n -- N ) 1- for step next ;
net 5 steps img forward lbl loss.ce ." warm_ce " .
variable t0 clock t0 !
100 steps img forward lbl loss.ce clock t0 @ - ." ms_for_100 " . ." ce " .
bye
