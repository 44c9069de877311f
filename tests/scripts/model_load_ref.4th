\ f-2 interop: ref_model_roundtrip.t4 was written by the REFERENCE's own saver (src/io/aio_model.cpp:16-61,143-180, run on the CPU by
\ oracle/_ref/ten4_refhost, tools/regen_vm_goldens.py; committed under tests/golden/refhost/).  A freshly built net with other weights
\ loads it and must reproduce the forward output the reference's VM printed for the saved model (out1 of model_save_load.4th).
0 trace
4 12 12 1 nn.model 0.5 4 conv2d 2 maxpool relu flatten 6 linear softmax constant net
4 12 12 1 tensor rand constant img
net img forward ." before " -1 n@ sum . drop
s" ref_model_roundtrip.t4" load
img forward ." out " -1 n@ .
." w0 " 0 nn.w .
." b0 " 0 nn.b .
." w4 " 4 nn.w sum . drop
bye
