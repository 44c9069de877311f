\ the printer's forms (aio.cpp:38-57, aio_tensor.cpp:141-226, debug.cpp:64-81): scalars of every size and sign in base 10 and 16, field widths, vectors and
\ matrices at and beyond the elision thresholds, tensors with channels, views, non-finite values - golden from the reference's VM
0 trace
0 . 1 . -1 . 7.5 . -7.5 . 0.1 . 1000000 . 123456789 . 10000000000 . 0.000001 . 0.0000001 . -0.000002 . 3.14159265 .
1 3 / . 2 3 / . 100 3 / . -100 3 / .
hex 255 . -255 . 4096 . 0.5 . decimal
5 3 .r 12345 3 .r -7 6 .r 42 5 u.r cr
1 0 / . -1 0 / . 0 0 / .
hex 10 20 -30 .s decimal 2drop drop
10 vector ones .
11 vector gradfill .
3 vector{ -0 0.00001 -0.00001 } .
10 10 matrix gradfill .
11 3 matrix ones .
3 11 matrix gradfill .
12 12 matrix ones 0.25 *= .
1 2 2 3 tensor ={ 1 2 3 4 5 6 7 8 9 10 11 12 } .
2 1 2 1 tensor ={ 1 2 3 4 } .
2 12 1 2 tensor ones .
4 vector{ 1 2 3 4 } dup dup .s
drop drop drop
2 2 matrix{ 1 0 0 1 } 1 0 / *= .
1000 vector ones 0.001 *= dup sum . dup avg . dup max . min .
bye
