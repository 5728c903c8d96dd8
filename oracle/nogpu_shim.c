/* ORACLE / bench infrastructure -- not part of the product.
 *
 * Preloaded (LD_PRELOAD) into the single-threaded CPU-baseline worker processes of bench.py ONLY.  PyTorch's autograd engine
 * asks every registered backend for its device count on the first backward() call; on a ROCm build that initialises the HIP
 * runtime and opens the GPU device node although the worker computes on the CPU.  The GPU box admits only a few processes
 * with the device open, and a CPU baseline has no business touching it: this shim answers "no devices" without
 * initialising anything, so one worker per host core can run.  Nothing in the product or in the GPU path loads it. */
int hipGetDeviceCount(int *count) {
    if (count) *count = 0;
    return 100; /* hipErrorNoDevice */
}
/* c10::cuda::device_count() clears the error state after the failed query: also answered here (every HIP entry point
 * initialises the runtime on first use, hipGetLastError included) */
int hipGetLastError(void) { return 0; }
int hipPeekAtLastError(void) { return 0; }
