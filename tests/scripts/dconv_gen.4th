\ transposed-convolution generator (DCGAN style): two `dconv2d` layers (4x4, stride 2, padding 1) double the grid twice;
\ forward / MSE loss / backprop / Adam, then a save - load round trip of the dconv2d weights
0 trace
6 4 4 12 nn.model 0.5 8 dconv2d relu 0.5 2 dconv2d tanh constant gen
gen network
6 4 4 12 tensor randn constant z
6 16 16 2 tensor rand constant tgt
z forward ." mid " 1 n@ sum . drop ." out " -1 n@ sum . drop
tgt loss.mse ." mse " .
tgt backprop
." g_b0 " 0 nn.db .
." g_w0 " 0 nn.dw sum . drop ." g_b2 " 2 nn.db .
." g_w2 " 2 nn.dw sum . drop ." dz " 0 n@ sum . drop
0.01 nn.adam
." w2 " 2 nn.w sum . drop ." b0 " 0 nn.b .
z forward tgt loss.mse ." mse2 " .
tgt backprop 0.01 nn.adam z forward tgt loss.mse ." mse3 " .
drop
\ odd input grid: output padding 1 (7 -> 15)
2 7 7 3 nn.model 0.5 4 dconv2d constant odd
odd network
2 7 7 3 tensor rand constant zo
zo forward ." odd_out " -1 n@ sum . drop
2 15 15 4 tensor ones backprop ." odd_dx " 0 n@ sum . drop ." odd_db " 0 nn.db .
drop
bye
