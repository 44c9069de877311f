\ layer kinds no other script runs inside a model: logsoftmax + loss.nll, avgpool / minpool, selu / elu / leakyrelu / tanh
\ layers, `broadcast` of a [N,1] target, upsample; one forward / backprop / SGD step each
0 trace
4 12 12 2 nn.model
0.5 6 conv2d 2 avgpool selu
0.5 8 conv2d 2 minpool 1.0 elu
flatten 12 linear tanh 5 linear logsoftmax
constant net
net network
4 12 12 2 tensor rand constant img
20 vector zeros 1 2 t! 1 5 t! 1 14 t! 1 16 t! 4 1 5 1 reshape4 constant lbl
img forward ." logp " -1 n@ .
lbl loss.nll ." nll " .
lbl backprop
." g_conv0_b " 0 nn.db .
." g_conv3_b " 3 nn.db .
." g_lin7_w " 7 nn.dw sum . drop
." dx " 0 n@ sum . drop
0.05 0.0 nn.sgd
." w0 " 0 nn.w sum . drop ." w9 " 9 nn.w .
img forward ." logp2 " -1 n@ .
drop
\ regression head: sigmoid output, target given as [N,1] and broadcast over the output width
4 1 6 1 nn.model 3 linear 0.1 leakyrelu 2 linear sigmoid constant reg
4 1 6 1 tensor randn constant x
4 vector{ 1 0 1 0 } constant t1
reg x forward t1 broadcast ." hot " nn.onehot . 
backprop ." reg_db " 2 nn.db .
." reg_dw0 " 0 nn.dw .
drop
\ upsample (nearest) in front of a conv: forward shape and values, backward gradient
2 4 4 1 nn.model 2 upsample 0.5 2 conv2d constant up
up network
2 4 4 1 tensor rand constant ux
ux forward ." up_mid " 1 n@ sum . drop ." up_out " -1 n@ sum . drop
2 8 8 2 tensor ones backprop ." up_dx " 0 n@ .
drop
bye
