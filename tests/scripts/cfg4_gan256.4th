\ BASELINE config #4 at full size: the GAN nets of the reference's t4_40b demo (D 784-512-256-1 with leakyrelu + dropout +
\ sigmoid, G 128-256-512-784 with leakyrelu + tanh), N = 256, BCE, Adam beta1 = 0.5; two `train_d train_g` rounds on an
\ HBM-resident "real" batch, then both nets' parameters
0 trace
256 constant N
N 1 1 1 tensor ones  constant REAL
N 1 1 1 tensor zeros constant FAKE
N 28 28 1 nn.model 512 linear 0.2 leakyrelu 0.3 dropout 256 linear 0.2 leakyrelu 0.3 dropout 1 linear sigmoid constant D
N 128 1 1 nn.model 256 linear 0.2 leakyrelu 512 linear 0.2 leakyrelu 784 linear tanh constant G
N 28 28 1 tensor rand constant real
N 128 1 1 tensor randn constant Z
: F ( -- t4 ) G Z forward -1 n@ N 28 28 1 reshape4 swap drop ;
: train_d ( D -- D ) 1 trainable real forward REAL loss.bce . REAL backprop F forward FAKE loss.bce . FAKE backprop 0.0001 0.5 nn.adam ;
: train_g ( D -- D ) 0 trainable F forward REAL loss.bce . REAL backprop 0 n@ G swap backprop 0.0004 0.5 nn.adam drop ;
: rounds ( D n -- D ) 1- for train_d train_g cr next ;
D 2 rounds
." d_w0 " 0 nn.w sum . drop ." d_b0 " 0 nn.b sum . drop
." d_w3 " 3 nn.w sum . drop ." d_w6 " 6 nn.w sum . drop ." d_b6 " 6 nn.b .
." d_mask2 " 2 nn.ex sum . drop ." d_mask5 " 5 nn.ex sum . drop
real forward REAL loss.bce ." loss_real " .
drop
G ." g_w0 " 0 nn.w sum . drop ." g_w2 " 2 nn.w sum . drop ." g_w4 " 4 nn.w sum . drop ." g_b4 " 4 nn.b sum . drop
Z forward ." g_out " -1 n@ sum . drop
drop
bye
