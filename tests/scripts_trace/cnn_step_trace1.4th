\ LeNet-style CNN (conv-pool-relu x2, two linears, softmax): one forward / loss / backprop / SGD + Adam step   (trace level 1: the reference prints the input preview, a line per layer and the loss derivative)
1 trace
4 28 28 1 nn.model
0.5 10 conv2d 2 maxpool relu
0.5 20 conv2d 2 maxpool relu
flatten 100 linear 10 linear softmax
constant net
net network
4 28 28 1 tensor rand constant img
40 vector zeros 1 3 t! 1 15 t! 1 20 t! 1 39 t! 4 1 10 1 reshape4 constant lbl
img forward
." probs " -1 n@ .
lbl loss.ce ." ce " .
lbl backprop
." g_conv0_b " 0 nn.db .
." g_conv3_b " 3 nn.db .
." g_lin8_b " 8 nn.db .
." g_conv0_w " 0 nn.dw sum . drop
." g_conv3_w " 3 nn.dw sum . drop
." g_lin7_w " 7 nn.dw sum . drop
." dx_in " 0 n@ sum . drop
0.01 0.0 nn.sgd
." w0 " 0 nn.w .
." b8 " 8 nn.b .
img forward lbl backprop 0.01 0.0 nn.sgd
img forward ." probs2 " -1 n@ .
drop
\ Adam on a model of its own: the reference sizes the moment tensors at a model's FIRST optimizer step (gradient.cu:87), so one model keeps one optimizer
4 28 28 1 nn.model
0.5 10 conv2d 2 maxpool relu
flatten 10 linear softmax
constant net2
net2 img forward lbl backprop 0.001 nn.adam
." w0a " 0 nn.w .
img forward lbl backprop 0.001 nn.adam
img forward ." probs3 " -1 n@ .
bye
