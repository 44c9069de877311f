\ error paths: what the words say (and leave on the stack) when their operands are wrong - unknown words, shape mismatches, non-square matrices, a model word
\ without a model, layers that do not fit.  The golden is the reference VM's own output (oracle/_ref/ten4_refhost).
0 trace
foo
2 3 matrix ones 2 3 matrix ones matmul
2drop
2 3 matrix ones inverse
drop
2 3 matrix ones 3 3 matrix ones +
2drop
5 vector ones 4 vector ones +
2drop
1 2 matmul
2drop
forward
backprop
3 network
drop
2 2 matrix ones nn.w
drop
5 nn.sgd
drop
2 4 4 1 nn.model 0.5 3 conv2d
9 linear linear
drop
2 4 4 1 nn.model 7 conv2d
drop
6 vector ones 4 2 reshape2
drop
2 2 matrix ones det
2drop
2 3 matrix ones det
2drop
3 vector ones transpose
drop
2 2 matrix ones 1 1 t!
.
1 0 /
.
-1 sqrt . 0 ln . 0 1/x .
bye
