/*
 * A HIP runtime that launches nothing: the handful of entry points libgpd.so's HOST code calls, for the host-side sanitizer build
 * (tests/test_host_sanitizers.py: the four units compiled `--cuda-host-only -fsanitize=address,undefined`, linked against this file
 * instead of libamdhip64).  Kernel launches are counted and their geometry recorded, so the test can also hold the launch arithmetic
 * (grid and block sizes, dynamic LDS) against what the kernels expect -- on a machine without a GPU.  Test infrastructure.
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned x, y, z; } dim3_;
static __thread dim3_ t_grid, t_block;
static __thread size_t t_shmem;
static __thread void* t_stream;
static int g_launches;
static unsigned g_last[7];          /* grid xyz | block xyz | dynamic LDS bytes */

int __hipPushCallConfiguration(dim3_ grid, dim3_ block, size_t shmem, void* stream) {
    t_grid = grid; t_block = block; t_shmem = shmem; t_stream = stream;
    return 0;
}
int __hipPopCallConfiguration(dim3_* grid, dim3_* block, size_t* shmem, void** stream) {
    *grid = t_grid; *block = t_block; *shmem = t_shmem; *stream = t_stream;
    return 0;
}
void** __hipRegisterFatBinary(const void* data) { static void* handle; (void)data; return &handle; }
void __hipRegisterFunction(void** modules, const void* host_fn, char* dev_fn, const char* dev_name, unsigned limit, void* tid, void* bid,
                           void* bdim, void* gdim, int* wsize) {
    (void)modules; (void)host_fn; (void)dev_fn; (void)dev_name; (void)limit; (void)tid; (void)bid; (void)bdim; (void)gdim; (void)wsize;
}
void __hipRegisterVar(void** modules, void* var, char* host_name, char* dev_name, int ext, size_t size, int constant, int global) {
    (void)modules; (void)var; (void)host_name; (void)dev_name; (void)ext; (void)size; (void)constant; (void)global;
}
void __hipUnregisterFatBinary(void** modules) { (void)modules; }

int hipLaunchKernel(const void* fn, dim3_ grid, dim3_ block, void** args, size_t shmem, void* stream) {
    (void)fn; (void)args; (void)stream;
    ++g_launches;
    g_last[0] = grid.x; g_last[1] = grid.y; g_last[2] = grid.z; g_last[3] = block.x; g_last[4] = block.y; g_last[5] = block.z;
    g_last[6] = (unsigned)shmem;
    return 0;
}
int hipGetLastError(void) { return 0; }
const char* hipGetErrorString(int e) { return e ? "stub error" : "no error"; }
int hipGetDevice(int* dev) { *dev = 0; return 0; }
int hipDeviceGetAttribute(int* value, int attr, int dev) { (void)attr; (void)dev; *value = 100000; return 0; }
int hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
int hipFree(void* p) { free(p); return 0; }
int hipHostMalloc(void** p, size_t n, unsigned flags) { (void)flags; *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
int hipMemcpyAsync(void* dst, const void* src, size_t n, int kind, void* stream) { (void)kind; (void)stream; memcpy(dst, src, n); return 0; }
int hipStreamSynchronize(void* stream) { (void)stream; return 0; }
int hipMemcpyFromSymbolAsync(void* dst, const void* sym, size_t n, size_t off, int kind, void* stream) {
    (void)sym; (void)off; (void)kind; (void)stream; memset(dst, 0, n); return 0;
}
int hipMemcpyToSymbolAsync(const void* sym, const void* src, size_t n, size_t off, int kind, void* stream) {
    (void)sym; (void)src; (void)n; (void)off; (void)kind; (void)stream; return 0;
}

/* what the test reads */
int hipstub_launches(void) { return g_launches; }
void hipstub_last(unsigned out[7]) { memcpy(out, g_last, sizeof g_last); }
