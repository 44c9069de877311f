/* ten4.h - embedding API of the host VM (libten4.so): run Forth source in-process.
 *
 * The reference is a stand-alone REPL (src/ten4.cu:224-235 feeds stdin lines to the VM); this is the
 * same outer interpreter behind four C functions, so a launcher that owns the process (one rank per
 * GPU under torch.distributed) can drive the VM and get at the one buffer data-parallel training
 * exchanges - the gradient slab.  Plain C ABI, no C++ or torch types.
 */
#ifndef TEN4_H
#define TEN4_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ten4_vm ten4_vm;

/* create a VM (t4k_init on device T4_DEVICE / `device` if >= 0, seed `seed`); NULL when no GPU */
ten4_vm *ten4_new(int device, unsigned long long seed, int trace_level);
void ten4_free(ten4_vm *vm);
/* evaluate Forth source (any number of lines); returns 1 while the VM runs, 0 after `bye` */
int ten4_eval(ten4_vm *vm, const char *source);
/* text printed since the last call (valid until the next ten4_* call on this VM) */
const char *ten4_output(ten4_vm *vm);
/* gradient slab of the model that ran `forward`/`backprop` last: all dW|dB back to back.
 * Returns 0 and a device pointer + float count, or -1 when no model has been finalized. */
int ten4_grad_slab(ten4_vm *vm, float **dev_ptr, long *n_floats);
/* the stream every kernel of the VM is issued on (a hipStream_t) */
void *ten4_stream(ten4_vm *vm);
/* Each VM owns a Philox stream (seed given to ten4_new, position 0): ten4_eval swaps it into the backend and saves it back, so
 * VMs created with the same seed are identical replicas whatever the interleaving of their ten4_eval calls.  tell / seek read and
 * move the position (in elements) between evals - a data-parallel launcher that draws only its shard of a replicated tensor. */
unsigned long long ten4_rand_tell(ten4_vm *vm);
void ten4_rand_seek(ten4_vm *vm, unsigned long long element_offset);
/* NOTE for hosts that also call the kernel library directly: because ten4_eval installs the VM's own (seed, position) on entry, a
 * t4k_rand_init / t4k_rand_set_offset made between two evals is OVERRIDDEN by the next eval (and draws made through t4k_* entry points
 * between evals do not advance the VM's position).  To change the VM's stream use ten4_rand_reseed (new seed, position 0) or
 * ten4_rand_seek; the shard set by t4k_rand_set_shard is process-global and is not swapped. */
void ten4_rand_reseed(ten4_vm *vm, unsigned long long seed);
/* Copy the tensor on top of the data stack to host memory as fp32 (synchronises the VM stream).  Returns its element count (-1
 * when the top of stack is not a tensor); nothing is copied when cap < count or dst is NULL; shape = {H, W, C, N} if non-NULL.
 * Full-precision read-back for hosts and tests - the printer rounds to 4 decimals, `bin save` to 8 bits (aio_tensor.cpp:240-255). */
long ten4_fetch(ten4_vm *vm, float *dst, long cap, int shape[4]);
/* the inverse: fill the tensor on top of the data stack from n host floats (n must equal its element count; returns n or -1) */
long ten4_store(ten4_vm *vm, const float *src, long n);
/* Called during `backprop`, on the calling thread, right after the kernels that complete one layer's dW|dB have been
 * enqueued on the VM stream: `off`/`n` locate that layer's segment in the gradient slab (floats).  Layers finish in
 * reverse order, so [off, slab end) is complete (stream-ordered) at each call - a data-parallel launcher can start
 * all-reducing the tail of the slab while the earlier layers are still back-propagating.  NULL disables. */
typedef void (*ten4_grad_hook_fn)(int layer, long off, long n, void *user);
void ten4_set_grad_hook(ten4_vm *vm, ten4_grad_hook_fn fn, void *user);
/* Launch plan switch of the process (the run-time form of T4_LAZY_DX0): 1 = the first layer's dX is produced when a word reads it (default), 0 = stored by every
 * backprop as the reference does.  Read at every backprop; returns the previous setting.  bench.py times its headline step with 0, its other legs with 1. */
int ten4_set_lazy_dx0(int on);

#ifdef __cplusplus
}
#endif
#endif
