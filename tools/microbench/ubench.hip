// Microbenchmarks for gfx950 (MI355X): (1) HBM streaming reads under the access patterns of the tiled passes, (2) VALU issue
// rates of the instructions the screening kernels are made of.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip
// Run on the GPU box: ./ubench [MB]   -> one line per experiment.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T, int NB>
__global__ void k_stream_grid(const T* __restrict__ p, size_t n, unsigned long long* out)
{
    unsigned long long acc = 0;
    const size_t nth = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (NB - 1) * nth < n; i += NB * nth) {
        T v[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) v[q] = p[i + q * nth];
#pragma unroll
        for (int q = 0; q < NB; q++) acc += ((const unsigned*)&v[q])[0];
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

// persistent workgroups over items of `item` elements, dynamic item fetch, NB-deep double-buffered batches (the tiled passes)
template <class T, int NB>
__global__ void k_stream_items(const T* __restrict__ p, size_t n, int item, int* cursor, unsigned long long* out)
{
    __shared__ int next;
    unsigned long long acc = 0;
    const int n_items = (int)(n / item);
    const int nth = blockDim.x;
    for (int it = blockIdx.x; it < n_items;) {
        const T* src = p + (size_t)it * item;
        T nx[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) nx[q] = src[min((int)threadIdx.x + q * nth, item - 1)];
        __syncthreads();
        if (threadIdx.x == 0) next = gridDim.x + atomicAdd(cursor, 1);
        for (int e0 = threadIdx.x; e0 < item; e0 += NB * nth) {
            T v[NB];
#pragma unroll
            for (int q = 0; q < NB; q++) v[q] = nx[q];
#pragma unroll
            for (int q = 0; q < NB; q++) nx[q] = src[min(e0 + (NB + q) * nth, item - 1)];
#pragma unroll
            for (int q = 0; q < NB; q++) acc += ((const unsigned*)&v[q])[0];
        }
        __syncthreads();
        it = next;
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

template <int KIND>
__global__ void k_valu(float* out, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 0.1f, a2 = a0 + 0.2f, a3 = a0 + 0.3f, a4 = a0 + 0.4f, a5 = a0 + 0.5f, a6 = a0 + 0.6f, a7 = a0 + 0.7f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float b = 0.999f, c = 1e-3f;
    const f2 pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) { // 8 independent v_fma_f32 chains
                a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
                a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
            } else if (KIND == 1) { // 4 independent v_pk_fma_f32 chains (8 fma per 4 instructions)
                p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc);
                p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
            } else if (KIND == 2) { // v_log_f32
                a0 = __builtin_amdgcn_logf(a0) + 3.0f; a1 = __builtin_amdgcn_logf(a1) + 3.0f; a2 = __builtin_amdgcn_logf(a2) + 3.0f; a3 = __builtin_amdgcn_logf(a3) + 3.0f;
                a4 = __builtin_amdgcn_logf(a4) + 3.0f; a5 = __builtin_amdgcn_logf(a5) + 3.0f; a6 = __builtin_amdgcn_logf(a6) + 3.0f; a7 = __builtin_amdgcn_logf(a7) + 3.0f;
            } else if (KIND == 3) { // v_exp_f32
                a0 = __builtin_amdgcn_exp2f(a0) * 0.25f; a1 = __builtin_amdgcn_exp2f(a1) * 0.25f; a2 = __builtin_amdgcn_exp2f(a2) * 0.25f; a3 = __builtin_amdgcn_exp2f(a3) * 0.25f;
                a4 = __builtin_amdgcn_exp2f(a4) * 0.25f; a5 = __builtin_amdgcn_exp2f(a5) * 0.25f; a6 = __builtin_amdgcn_exp2f(a6) * 0.25f; a7 = __builtin_amdgcn_exp2f(a7) * 0.25f;
            } else if (KIND == 4) { // v_fma_f64
                double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
                d0 = __builtin_fma(d0, 0.999, 1e-3); d1 = __builtin_fma(d1, 0.999, 1e-3); d2 = __builtin_fma(d2, 0.999, 1e-3); d3 = __builtin_fma(d3, 0.999, 1e-3);
                d0 = __builtin_fma(d0, 0.999, 1e-3); d1 = __builtin_fma(d1, 0.999, 1e-3); d2 = __builtin_fma(d2, 0.999, 1e-3); d3 = __builtin_fma(d3, 0.999, 1e-3);
                a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <class F>
static float time_ms(F f, int reps = 5)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    return best;
}

// (3) no-return 64-bit atomic adds into a table of `bins` words: every lane its own pseudo-random bin (a histogram update),
// `per` adds per thread.  spread: the bins a wave touches lie within `spread` consecutive words (0: anywhere)
__global__ void k_atomics(unsigned long long* tab, unsigned bins, int per, unsigned spread)
{
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    const unsigned base = spread ? ((blockIdx.x * 97u + (threadIdx.x >> 6)) * 7919u) % (bins - spread) : 0u;
    for (int i = 0; i < per; i++) {
        x = x * 1664525u + 1013904223u;
        const unsigned b = spread ? base + (x >> 8) % spread : (x >> 8) % bins;
        __hip_atomic_fetch_add(&tab[b], (unsigned long long)(x & 1023u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// (4) a chain of dependent launches: a kernel of `wgs` workgroups that each spin for ~`ns` (s_memrealtime: 100 MHz), launched
// back to back on one stream, and the same chain as the kernel nodes of a hipGraph (instantiated once, launched again and again)
__global__ void k_spin(unsigned long long* out, int ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0;
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 400;
    const size_t bytes = mb << 20;
    void* buf;
    unsigned long long* out;
    int* cursor;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&cursor, 4));
    CK(hipMemset(buf, 1, bytes));
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s, %d CUs, clock %d MHz\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000);
    auto report = [&](const char* name, float ms) { printf("%-64s %8.1f us  %7.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); };
    report("grid-stride  8-byte loads, 4 per thread, 4096 x 256", time_ms([&] { hipLaunchKernelGGL((k_stream_grid<uint2, 4>), dim3(4096), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, out); }));
    report("grid-stride 16-byte loads, 4 per thread, 4096 x 256", time_ms([&] { hipLaunchKernelGGL((k_stream_grid<uint4, 4>), dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out); }));
    report("grid-stride 16-byte loads, 8 per thread, 2048 x 256", time_ms([&] { hipLaunchKernelGGL((k_stream_grid<uint4, 8>), dim3(2048), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out); }));
    report("grid-stride  8-byte loads, 8 per thread, 512 x 512 (2 per CU)", time_ms([&] { hipLaunchKernelGGL((k_stream_grid<uint2, 8>), dim3(512), dim3(512), 0, 0, (const uint2*)buf, bytes / 8, out); }));
    report("grid-stride 16-byte loads, 8 per thread, 512 x 512 (2 per CU)", time_ms([&] { hipLaunchKernelGGL((k_stream_grid<uint4, 8>), dim3(512), dim3(512), 0, 0, (const uint4*)buf, bytes / 16, out); }));
    report("grid-stride 16-byte loads, 8 per thread, 1024 x 512 (4 per CU)", time_ms([&] { hipLaunchKernelGGL((k_stream_grid<uint4, 8>), dim3(1024), dim3(512), 0, 0, (const uint4*)buf, bytes / 16, out); }));
    auto items = [&](auto kern, int grid, int threads, int item_elems, size_t n) {
        return time_ms([&] {
            CK(hipMemsetAsync(cursor, 0, 4));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, (decltype(buf))buf, n, item_elems, cursor, out);
        });
    };
    (void)items;
    report("items of 16384 x 8 B, persistent 512 x 512, 8-deep", time_ms([&] { CK(hipMemsetAsync(cursor, 0, 4)); hipLaunchKernelGGL((k_stream_items<uint2, 8>), dim3(512), dim3(512), 0, 0, (const uint2*)buf, bytes / 8, 16384, cursor, out); }));
    report("items of 16384 x 8 B, persistent 512 x 1024, 2-deep", time_ms([&] { CK(hipMemsetAsync(cursor, 0, 4)); hipLaunchKernelGGL((k_stream_items<uint2, 2>), dim3(512), dim3(1024), 0, 0, (const uint2*)buf, bytes / 8, 16384, cursor, out); }));
    report("items of 8192 x 16 B, persistent 512 x 512, 8-deep", time_ms([&] { CK(hipMemsetAsync(cursor, 0, 4)); hipLaunchKernelGGL((k_stream_items<uint4, 8>), dim3(512), dim3(512), 0, 0, (const uint4*)buf, bytes / 16, 8192, cursor, out); }));
    report("items of 8192 x 16 B, persistent 1024 x 256, 8-deep", time_ms([&] { CK(hipMemsetAsync(cursor, 0, 4)); hipLaunchKernelGGL((k_stream_items<uint4, 8>), dim3(1024), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, 8192, cursor, out); }));
    report("items of 8192 x 16 B, persistent 2048 x 256, 4-deep", time_ms([&] { CK(hipMemsetAsync(cursor, 0, 4)); hipLaunchKernelGGL((k_stream_items<uint4, 4>), dim3(2048), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, 8192, cursor, out); }));
    // the same on a buffer that fits the Infinity Cache (160 MB: what a pass reads)
    {
        const size_t b2 = (size_t)160 << 20;
        auto rep2 = [&](const char* name, float ms) { printf("%-64s %8.1f us  %7.2f TB/s\n", name, ms * 1e3, b2 / (ms * 1e-3) / 1e12); };
        rep2("160 MB again and again: grid-stride 16-byte, 2048 x 256", time_ms([&] { hipLaunchKernelGGL((k_stream_grid<uint4, 8>), dim3(2048), dim3(256), 0, 0, (const uint4*)buf, b2 / 16, out); }, 10));
        rep2("160 MB again and again: items 16384 x 8 B, 512 x 512, 8-deep", time_ms([&] { CK(hipMemsetAsync(cursor, 0, 4)); hipLaunchKernelGGL((k_stream_items<uint2, 8>), dim3(512), dim3(512), 0, 0, (const uint2*)buf, b2 / 8, 16384, cursor, out); }, 10));
    }
    // ---- VALU issue rates: 256 CUs x 4 SIMDs, 8 waves per SIMD, 8 independent chains per lane
    float* fo;
    const int grid = pr.multiProcessorCount * 8, threads = 256, iters = 4096;
    CK(hipMalloc(&fo, (size_t)grid * threads * 4));
    const double lanes = (double)grid * threads;
    auto valu = [&](const char* name, float ms, double ops_per_iter) {
        const double rate = lanes * iters * ops_per_iter / (ms * 1e-3);
        printf("%-64s %8.1f us  %7.2f T lane-ops/s  (%.1f lane-ops per CU per clock at %d MHz)\n", name, ms * 1e3, rate / 1e12,
               rate / pr.multiProcessorCount / (pr.clockRate * 1e3), pr.clockRate / 1000);
    };
    valu("v_fma_f32 (64 per iteration, 8 chains)", time_ms([&] { hipLaunchKernelGGL(k_valu<0>, dim3(grid), dim3(threads), 0, 0, fo, iters); }), 64);
    valu("v_pk_fma_f32 (32 instructions = 64 fma per iteration)", time_ms([&] { hipLaunchKernelGGL(k_valu<1>, dim3(grid), dim3(threads), 0, 0, fo, iters); }), 64);
    valu("v_log_f32 + v_add_f32 (64 + 64 per iteration)", time_ms([&] { hipLaunchKernelGGL(k_valu<2>, dim3(grid), dim3(threads), 0, 0, fo, iters); }), 64);
    valu("v_exp_f32 + v_mul_f32 (64 + 64 per iteration)", time_ms([&] { hipLaunchKernelGGL(k_valu<3>, dim3(grid), dim3(threads), 0, 0, fo, iters); }), 64);
    valu("v_fma_f64 (64 per iteration, + 8 x 8 conversions)", time_ms([&] { hipLaunchKernelGGL(k_valu<4>, dim3(grid), dim3(threads), 0, 0, fo, iters); }), 64);
    // ---- (3) atomics
    {
        unsigned long long* tab;
        const unsigned bins = 3 * 49152;
        CK(hipMalloc(&tab, bins * 8));
        CK(hipMemset(tab, 0, bins * 8));
        auto atom = [&](const char* name, int g, int t, int per, unsigned spread) {
            const float ms = time_ms([&] { hipLaunchKernelGGL(k_atomics, dim3(g), dim3(t), 0, 0, tab, bins, per, spread); });
            const double n = (double)g * t * per;
            printf("%-64s %8.1f us  %7.2f G atomics/s\n", name, ms * 1e3, n / (ms * 1e-3) / 1e9);
        };
        atom("64-bit atomic add, 147 k bins, 1 M adds (256 x 256 x 16)", 256, 256, 16, 0);
        atom("64-bit atomic add, 147 k bins, 16 M adds (2048 x 256 x 32)", 2048, 256, 32, 0);
        atom("64-bit atomic add, 147 k bins, 64 M adds (4096 x 256 x 64)", 4096, 256, 64, 0);
        atom("... a wave's bins within 256 words, 16 M adds", 2048, 256, 32, 256);
        atom("... a wave's bins within 16 words, 16 M adds", 2048, 256, 32, 16);
        atom("64-bit atomic add, 147 k bins, 64 k adds (64 x 256 x 4)", 64, 256, 4, 0);
    }
    // ---- (4) launch chains: stream against hipGraph
    {
        hipStream_t st;
        CK(hipStreamCreate(&st));
        unsigned long long* o2;
        CK(hipMalloc(&o2, 64));
        for (int wgs : {1, 2048}) {
            for (int us : {5, 30}) {
                const int n = 13, ticks = us * 100;
                auto chain = [&] { for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, st, o2, ticks); };
                auto run_ms = [&](auto f) {
                    hipEvent_t a, b;
                    CK(hipEventCreate(&a));
                    CK(hipEventCreate(&b));
                    f();
                    CK(hipStreamSynchronize(st));
                    float best = 1e30f;
                    for (int r = 0; r < 20; r++) {
                        CK(hipEventRecord(a, st));
                        f();
                        CK(hipEventRecord(b, st));
                        CK(hipEventSynchronize(b));
                        float ms;
                        CK(hipEventElapsedTime(&ms, a, b));
                        best = ms < best ? ms : best;
                    }
                    return best;
                };
                const float ms_stream = run_ms(chain);
                hipGraph_t gr;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
                chain();
                CK(hipStreamEndCapture(st, &gr));
                CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
                const float ms_graph = run_ms([&] { CK(hipGraphLaunch(ge, st)); });
                printf("13 dependent launches of %4d workgroups x %2d us: stream %7.1f us (%.1f us per launch beyond its work), hipGraph %7.1f us (%.1f)\n", wgs,
                       us, ms_stream * 1e3, (ms_stream * 1e3 - n * us) / n, ms_graph * 1e3, (ms_graph * 1e3 - n * us) / n);
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(gr));
            }
        }
    }
    return 0;
}
