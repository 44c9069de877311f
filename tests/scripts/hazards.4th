\ decided reference hazards (SURVEY.md 9, DESIGN.md 7): words whose code in the reference cannot do what its README documents.  The golden of this
\ script comes from the product's host over the oracle, not from the reference's VM (tools/regen_vm_goldens.py HAZARD).
0 trace
\ t@ ( T i -- T n ), README.md:548: the guard at tenvm.cpp:536 tests the operands the wrong way round, `T i t@` is a no-op there
6 vector{ 1 2 3 4 5 6 } 2 3 reshape2 4 t@ ." t@ " . 10 4 t! 4 t@ ." t! " .
drop
\ slice of a rank-4 tensor: mmu.cu:320-325 copies the window of sample 0 only (N times); every sample's window is copied here
2 2 2 1 tensor ={ 1 2 3 4 5 6 7 8 } 0 1 0 2 slice ." slice " .
drop
\ a second optimizer on one model: the moment tensors are sized at the model's first step only (gradient.cu:87), a later nn.adam
\ dereferences NULL there; here they are allocated at the first Adam step
2 1 4 1 nn.model 3 linear tanh 2 linear softmax constant net
2 1 4 1 tensor ={ 1 0 -1 0.5 0.25 -0.5 1 2 } constant x
4 vector{ 1 0 0 1 } 2 1 2 1 reshape4 constant y
net x forward y backprop 0.1 0.0 nn.sgd
x forward y backprop 0.01 nn.adam
x forward ." sgd_then_adam " -1 n@ .
drop
bye
