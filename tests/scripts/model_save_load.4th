\ train a small net, save it (.t4 model file), build a second identical net with different weights, load, compare outputs
0 trace
4 12 12 1 nn.model 0.5 4 conv2d 2 maxpool relu flatten 6 linear softmax constant net1
4 12 12 1 tensor rand constant img
24 vector zeros 1 1 t! 1 6 t! 1 14 t! 1 21 t! 4 1 6 1 reshape4 constant lbl
: step ( N -- N ) img forward lbl backprop 0.05 nn.adam ;
net1 step step step
img forward ." out1 " -1 n@ .
s" model_roundtrip.t4" save
drop
4 12 12 1 nn.model 0.5 4 conv2d 2 maxpool relu flatten 6 linear softmax constant net2
net2 img forward ." out2_before " -1 n@ sum .
drop
s" model_roundtrip.t4" load
img forward ." out2 " -1 n@ .
." w0 " 0 nn.w sum .
drop
drop
\ loading needs a built network, and one whose layers are those of the file: both mistakes are reported, not silently accepted
4 12 12 1 nn.model constant net3
net3 s" model_roundtrip.t4" load
drop
4 12 12 1 nn.model 0.5 4 conv2d 2 maxpool relu flatten 6 linear relu 3 linear softmax constant net4
net4 s" model_roundtrip.t4" load
drop
bye
