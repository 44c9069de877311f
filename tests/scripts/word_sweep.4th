\ every dictionary word no other script touches, once, on both VMs (the reference's own, oracle/_ref/ten4_refhost, prints the golden): stack, logic,
\ bit, number-format and compiler words, then the tensor and nn words that the model scripts leave out
0 trace
1 2 3 -rot .s 2drop drop
1 2 3 4 2over .s 2drop 2drop 2drop
1 2 3 4 2swap .s 2drop 2drop
5 ?dup 0 ?dup .s 2drop drop
1 2 nip . 7 8 9 2 pick . 2drop drop
-3 0< . 3 0< . -3 0> . 3 0> .
2 3 <= . 3 3 <= . 4 3 <= . 2 3 >= . 3 3 >= . 2 3 <> . 3 3 <> . 4 3 > . 2 3 > .
2 3 u< . 3 2 u< . 3 2 u> .
5 2* . 5 2/ . 7.5 f>s . 2.5 round . 2.4 ceil . 7 3 fmod . 7.5 2 fmod .
12 10 or . 12 10 xor . 1 3 lshift . 16 2 rshift . 0 invert . 5 invert .
65 emit space 66 emit bl emit 67 emit cr
255 u. -1 u. 3.7 u.
hex 255 . 255 u. decimal 255 .
base @ .
: cnt1 10 begin 1- dup 5 < if exit then again ; cnt1 .
: cnt2 0 10 for aft 1+ then next ; cnt2 .
: cnt3 0 10 0 do i + i 5 = if leave then loop ; cnt3 .
: sq dup * ; ' sq . 6 ' sq exec .
: dbl 2 * ; : op sq ; 5 op . ' dbl is op 5 op .
variable buf 16 allot 65 buf c! buf c@ . 7 buf 2 th ! buf 2 th @ . buf 8 + @ .
nop 1 abort .s
\ ---- compiled forms as `see` shows them (addresses, dictionary indices, branch targets: Debug::see debug.cpp:136-249), `here`, defining words
: w1 10 for i . next ;
: w2 5 0 do i . loop ;
: w3 ." hello " s" abc" 2drop ;
: w4 begin dup 0 > while 1- repeat ;
: w5 begin 1- dup 0 = until ;
variable v1 42 v1 !
7 constant c7
3 value x3
create arr 1 , 2 , 3 ,
: mk create , does> @ ;
5 mk five
see w1
see w2
see w3
see w4
see w5
see v1
see c7
see x3
see arr
see mk
see five
see dup
here .
five . arr @ . arr 2 th @ . 9 to x3 x3 . c7 . 3 w4 . w3 w1 w2
0.5 sin . 0.5 cos . 1 ms
words
\ ---- tensor words
3 3 matrix eye
.
2 2 matrix{ 1 2 3 4 } 2 2 matrix{ 5 6 7 8 } matmul
.
2drop
2 2 matrix{ 4 7 2 6 } 2 2 matrix{ 1 0 0 1 } matdiv
.
2drop
2 2 matrix{ 1 2 3 4 } copy 10 *= swap
.
.
2 3 matrix{ 1 2 3 4 5 6 } 2 3 matrix ones same_shape? . 2drop
2 3 matrix{ 1 2 3 4 5 6 } 3 2 matrix ones same_shape? . 2drop
4 vector{ 0.5 1 2 4 } dup 1/x
.
4 vector{ 1 10 100 1000 } dup log
.
4 vector{ -1 0.25 0.75 2 } dup sat
.
2 3 pow . 3 2 pow .
2 2 matrix{ 1 2 3 4 } dup 2 /=
.
2 2 matrix{ 1 2 3 4 } dup view .s 2drop drop
2.0 0.5 2 2 matrix{ 1 2 3 4 } 2 2 matrix{ 5 6 7 8 } 2 2 matrix ones gemm1
.
drop drop drop drop
2.0 0.5 2 2 matrix{ 1 2 3 4 } 2 2 matrix{ 5 6 7 8 } 2 2 matrix ones gemm2
.
drop drop drop drop
2.0 0.5 2 2 matrix{ 1 2 3 4 } 2 2 matrix{ 5 6 7 8 } 2 2 matrix ones gemm3
.
drop drop drop drop
2.0 0.5 2 2 matrix{ 1 2 3 4 } 2 2 matrix{ 5 6 7 8 } 2 2 matrix ones gemm4
.
drop drop drop drop
w/o . r/w . bin .
\ ---- nn words
2 6 6 1 nn.model 0.5 2 conv1x1 relu flatten 3 linear softmax constant net
net network
batchsize . drop
2 6 6 1 tensor ={ 0.1 0.2 0.3 0.4 0.5 0.6 0.7 0.8 0.9 1.0 0.9 0.8 0.7 0.6 0.5 0.4 0.3 0.2 0.1 0.2 0.3 0.4 0.5 0.6 0.7 0.8 0.9 1.0 0.9 0.8 0.7 0.6 0.5 0.4 0.3 0.2 0.1 0.2 0.3 0.4 0.5 0.6 0.7 0.8 0.9 1.0 0.9 0.8 0.7 0.6 0.5 0.4 0.3 0.2 0.1 0.2 0.3 0.4 0.5 0.6 0.7 0.8 0.9 1.0 0.9 0.8 0.7 0.6 0.5 0.4 0.3 0.2 } constant img
6 vector{ 1 0 0 0 0 1 } 2 1 3 1 reshape4 constant hot
net img forward ." out " -1 n@ .
hot loss.ce ." ce " . hot nn.loss ." nnloss " .
hot backprop ." db4 " 4 nn.db .
nn.zero ." zeroed " 4 nn.db .
1.5 nn.max_norm
drop
bye
