\ BASELINE config #4: GAN generator + discriminator training (t4_40b nets: D 784-512-256-1, G 128-256-512-784, N=256,
\ Adam b1=0.5), synthetic HBM-resident "real" batch; timed train_d + train_g rounds (the loss read-back in front of the second `clock` is the device sync: the VM enqueues
\ asynchronously, without it the figure is the host's enqueue time)
0 trace
256 constant N
N 1 1 1 tensor ones  constant REAL
N 1 1 1 tensor zeros constant FAKE
N 28 28 1 nn.model 512 linear 0.2 leakyrelu 0.3 dropout 256 linear 0.2 leakyrelu 0.3 dropout 1 linear sigmoid constant D
N 128 1 1 nn.model 256 linear 0.2 leakyrelu 512 linear 0.2 leakyrelu 784 linear tanh constant G
N 28 28 1 tensor rand constant real
N 128 1 1 tensor randn constant Z
: F ( -- t4 ) G Z forward -1 n@ N 28 28 1 reshape4 swap drop ;
: train_d ( D -- D ) 1 trainable real forward REAL backprop F forward FAKE backprop 0.0001 0.5 nn.adam ;
: train_g ( D -- D ) 0 trainable F forward REAL backprop 0 n@ G swap backprop 0.0004 0.5 nn.adam drop ;
: rounds ( D n -- D ) 1- for train_d train_g next ;
D 10 rounds real forward REAL loss.bce ." warm_loss_real " .
variable t0 clock t0 !
40 rounds real forward REAL loss.bce clock t0 @ - ." ms_for_40 " . ." loss_real " . F forward REAL loss.bce ." loss_gen " .
bye
