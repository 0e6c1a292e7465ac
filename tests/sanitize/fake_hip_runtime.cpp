// A HIP runtime that is not one: what libinstagraal_hip.so's HOST side needs from libamdhip64, implemented on the heap, so that
// the unmodified csrc/ig_hip.hip -- compiled with `hipcc --offload-host-only -fsanitize=address,undefined` -- links and runs on a
// machine without a GPU, under the sanitizers (tests/test_cpu_abi_and_host.py::test_host_logic_under_address_and_ub_sanitizers).
//   * "device" memory is malloc'ed memory: every hipMemcpy / hipMemset of the library is checked by AddressSanitizer against
//     the real allocation sizes;
//   * a kernel launch calls the MODEL registered for that kernel (tests/sanitize/host_logic_harness.cpp scripts the few kernels
//     whose outputs steer the host: the decide step's outcome, the records and flags written to mapped host memory) or does
//     nothing; streams are synchronous, events are trivially complete.
// Test infrastructure only: nothing here is part of the product.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "fake_hip_runtime.h"

namespace {
std::map<const void*, std::string>& names()
{
    static std::map<const void*, std::string> m;
    return m;
}
std::map<std::string, fake_hip::Model>& models()
{
    static std::map<std::string, fake_hip::Model> m;
    return m;
}
struct CallCfg {
    dim3 grid, block;
    size_t shmem;
    hipStream_t stream;
};
thread_local CallCfg g_cfg;
long g_launches = 0, g_allocs = 0, g_alloc_bytes = 0;
int g_fail_alloc_at = -1; // the n-th hipMalloc from now fails (error-path tests)
}  // namespace

namespace fake_hip {
void set_model(const std::string& kernel_substring, Model m) { models()[kernel_substring] = std::move(m); }
void clear_models() { models().clear(); }
long launches() { return g_launches; }
long allocations() { return g_allocs; }
void fail_allocation_in(int n) { g_fail_alloc_at = n; }
}  // namespace fake_hip

extern "C" {
// ---- registration (what the host-side stubs of the kernels call at load time)
void** __hipRegisterFatBinary(const void*)
{
    static void* handle = nullptr;
    return &handle;
}
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void* host_fn, char*, const char* device_name, unsigned, void*, void*, void*, void*, int*)
{
    names()[host_fn] = device_name ? device_name : "?";
}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    g_cfg = CallCfg{grid, block, shmem, stream};
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream)
{
    *grid = g_cfg.grid;
    *block = g_cfg.block;
    *shmem = g_cfg.shmem;
    *stream = g_cfg.stream;
    return hipSuccess;
}
}

hipError_t hipLaunchKernel(const void* fn, dim3 grid, dim3 block, void** args, size_t, hipStream_t)
{
    g_launches++;
    auto it = names().find(fn);
    const std::string name = it == names().end() ? std::string("?") : it->second;
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0 || block.x * block.y * block.z > 1024) {
        fprintf(stderr, "fake HIP: bad launch geometry for %s: grid %u %u %u block %u %u %u\n", name.c_str(), grid.x, grid.y, grid.z, block.x, block.y,
                block.z);
        abort();
    }
    for (auto& m : models())
        if (name.find(m.first) != std::string::npos) {
            m.second(args, grid, block);
            break;
        }
    return hipSuccess;
}

// ---- memory
hipError_t hipMalloc(void** p, size_t n)
{
    if (g_fail_alloc_at >= 0 && g_fail_alloc_at-- == 0) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    g_allocs++;
    g_alloc_bytes += (long)n;
    *p = calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p)
{
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned)
{
    *p = calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p)
{
    free(p);
    return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** dp, void* hp, unsigned)
{
    *dp = hp; // mapped memory: one address space here
    return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind)
{
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t)
{
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n)
{
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t)
{
    memset(d, v, n);
    return hipSuccess;
}

// ---- devices, streams, events: one device, everything synchronous
hipError_t hipGetDeviceCount(int* n)
{
    *n = 1;
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory (fake)" : "error (fake)"); }
hipError_t hipStreamCreate(hipStream_t* s)
{
    *s = (hipStream_t)malloc(8); // a handle that must be destroyed exactly once (LeakSanitizer / double free)
    return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest)
{
    *least = 0;
    *greatest = -1;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    free((void*)s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e)
{
    *e = (hipEvent_t)malloc(8);
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned)
{
    *e = (hipEvent_t)malloc(8);
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e)
{
    free((void*)e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t)
{
    *ms = 0.001f;
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
