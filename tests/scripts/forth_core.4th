\ scalar eForth words (no tensor kernels): arithmetic, logic, loops, defining words, strings
0 trace
1 2 3 rot .s drop drop drop
10 3 /mod . . 7 2 mod . -7 abs . 5 negate . 9 sqrt . 2 3 max . 2 3 min .
1949 1461 4 */mod . .
: sq dup * ; 7 sq .
: fact ( n -- n! ) 1 swap for r@ 1+ * next ; 5 fact .
variable v 42 v ! v @ . 8 v +! v ?
5 constant five five .
3 value vv vv . 9 to vv vv .
: tst 4 0 do i . loop ; tst
: ctr 3 for r@ . next ; ctr
: yes? if ." yes " else ." no " then ; 1 yes? 0 yes?
: cnt begin dup . 1- dup 0= until drop ; 3 cnt
: w5 0 begin dup 3 < while dup . 1+ repeat drop ; w5
create arr 1 , 2 , 3 , arr 2 cells + @ .
: mk create , does> @ ; 77 mk k77 k77 .
hex ff . decimal 255 .
$10 . %101 . #12 .
1.5 2.25 + . 1 3 / .
3 4 < . 4 3 < . 3 3 = . 0 0= .
s" hello" type cr
5 3 .r 7 4 u.r cr
depth .
bye
