\ matrix creation, matmul, element-wise words, reductions, transposes
0 trace
2 3 matrix{ 1 2 3 4 5 6 } dup .
3 2 matrix ones dup .
@ ." AB " .
2drop
2 3 matrix{ 1 2 3 4 5 6 } 2 3 matrix ones 2dup += ." sum " .
-= ." diff " .
2 3 matrix{ 1 2 3 0 4 5 } 3 2 matrix ones @= dup ." prod " .
2 2 matrix ones 0.5 *= *= ." had " .
3 4 matrix gradfill dup ." ramp " .
transpose ." rampT " .
drop
5 vector{ 1 2 3 4 5 } dup sum ." s " . dup avg ." a " . dup max ." mx " . dup min ." mn " . dup norm ." nrm " .
drop
4 vector{ 1 2 3 4 } 4 vector{ 4 3 2 1 } @ ." dot " .
2drop
20 vector ones 0.25 *= dup sum ." s20 " . std ." sd " .
drop
2.0 0.5 2 2 matrix{ 1 2 3 4 } 2 2 matrix{ 5 6 7 8 } 2 2 matrix ones gemm ." g " .
drop drop drop drop drop
4 vector{ 0.5 -1 2 -3 } dup relu ." relu " .
4 vector{ 0.5 -1 2 -3 } dup abs ." abs " .
4 vector{ 0 1 2 3 } dup exp ." exp " .
3 vector{ 1 2 3 } 2 * ." ts " .
drop
2 3 4 1 tensor ones dup dim ." dim " .
drop
6 vector{ 1 2 3 4 5 6 } 2 3 reshape2 dup ." rs " .
drop
bye
