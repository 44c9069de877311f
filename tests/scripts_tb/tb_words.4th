\ every TensorBoard word once (run with -t<logdir> -rrun1): scalar, histogram, text, image tile, model graph, embedding
0 trace
3 .tbstep
0.5 s" train/loss" .scalar
8 vector{ -1 0 0.25 0.5 1 2 2 3 } 4 s" nn/w" .histo
: note s" epoch three" s" notes" .text ;
note
7 .tbstep
2 2 3 1 tensor ={ 0 0.25 0.5 0.75 1 0.125 0.5 0.5 0.5 0.5 0.5 0.5 } 2 s" imgs" .tile
1.5 s" train/loss" .scalar
2 6 6 1 nn.model 0.5 2 conv2d 2 maxpool relu flatten 3 linear softmax constant net
net .graph
3 1 4 1 tensor ={ 1 2 3 4 0.5 0.25 0.125 0 -1 -2 -3 -4 } s" emb/z" .embed
bye
