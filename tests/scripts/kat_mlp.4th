\ two-layer sigmoid MLP with hand-set weights; numbers from tests/golden/kat_reference_examples.json (t4_30b)
0 trace
1 1 2 1 nn.model 3 linear sigmoid 2 linear sigmoid constant net
net
6 vector{ 0.15 0.2 0.25 0.3 0.2 0.15 } 0 nn.w=
3 vector{ 0.35 0.35 0.35 } 0 nn.b=
6 vector{ 0.4 0.45 0.5 0.55 0.5 0.45 } 2 nn.w=
2 vector{ 0.6 0.6 } 2 nn.b=
2 vector{ 0.05 0.1 } forward
." hidden_in " 1 n@ .
." hidden_mask " 1 nn.w .
." hidden_out " 2 n@ .
." out_in " 3 n@ .
." out " -1 n@ .
2 vector{ 0.01 0.99 } constant goal
goal loss.mse ." loss " .
goal backprop
." dy " 4 n@ .
." db2 " 2 nn.db .
." dw2 " 2 nn.dw .
." dx2 " 2 n@ .
." db0 " 0 nn.db .
." dw0 " 0 nn.dw .
." dx0 " 0 n@ .
0.5 0.0 nn.sgd
." w2 " 2 nn.w .
." b2 " 2 nn.b .
." w0 " 0 nn.w .
." b0 " 0 nn.b .
." dw2z " 2 nn.dw .
network
bye
