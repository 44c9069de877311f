\ small GAN on synthetic MNIST-shaped data: discriminator (leakyrelu + dropout + sigmoid, BCE) and generator (tanh)
\ trained in the alternating pattern of the reference's GAN demo, Adam with beta1 = 0.5
0 trace
16 constant N
N 1 1 1 tensor ones  constant REAL
N 1 1 1 tensor zeros constant FAKE
N 28 28 1 nn.model 64 linear 0.2 leakyrelu 0.3 dropout 32 linear 0.2 leakyrelu 1 linear sigmoid constant D
N 32 1 1 nn.model 64 linear 0.2 leakyrelu 784 linear tanh constant G
N dataset mnist_train
128 128 normalize
constant ds0
: Z N 32 1 1 tensor randn ;
: F ( -- t4 ) G Z forward -1 n@ N 28 28 1 reshape4 swap drop ;
: train_d ( D -- D ) 1 trainable
  ds0 forward REAL loss.bce . REAL backprop
  F forward FAKE loss.bce . FAKE backprop
  0.0002 0.5 nn.adam ;
: train_g ( D -- D ) 0 trainable
  F forward REAL loss.bce . REAL backprop
  0 n@ G swap backprop 0.0004 0.5 nn.adam drop ;
: rounds ( D n -- D ) 1- for train_d train_g cr next ;
D ds0 fetch
drop 3 rounds
." d_w0 " 0 nn.w sum . drop
." d_w5 " 5 nn.w .
drop
G ." g_w0 " 0 nn.w sum . drop
." g_b2 " 2 nn.b sum . drop
bye
