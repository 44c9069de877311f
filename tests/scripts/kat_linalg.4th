\ determinant, inverses, PLU round trip and solve; numbers from tests/golden (t4_22a)
0 trace
3 3 matrix{ 2 2 5 1 1 1 4 6 8 }
det ." det " .
inverse ." gj " .
luinv ." lu " .
drop
3 3 matrix{ 1 2 4 3 8 14 2 6 13 }
plu 2dup ." packed " .
." perm " .
lower dup ." L " .
swap
upper dup ." U " .
swap drop
@= @= ." PLU " .
." A " .
3 vector{ 1 1 1 } 3 3 matrix{ 5 7 4 3 -1 3 6 7 5 } solve dup ." x " .
@= ." Ax " .
." b " .
2 2 matrix{ 1 2 2 4 } inverse drop drop
4 4 matrix randn dup inverse @ ." MMinv " .
bye
