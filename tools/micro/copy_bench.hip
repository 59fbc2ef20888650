// copy_bench.hip -- what can a device copy of the compaction's size reach, and what does a dword-misaligned
// source cost?  5.3 GB int32 -> 5.3 GB (the k_compact workload of the bench matrix), forms:
//   A  aligned 16-B loads and stores, one vector per thread
//   B  aligned, grid-stride, U vectors in flight per thread
//   C  source displaced by 1 dword, 16-B loads at 4-byte alignment            [what k_compact issues]
//   D  source displaced by 1 dword, two ALIGNED 16-B loads + select            [candidate]
//   E  as C / D with non-temporal loads and stores
// build: hipcc --offload-arch=gfx950 -O3 -o copy_bench copy_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) V4U { int x, y, z, w; };

template <bool NT> __device__ __forceinline__ v4i ld(const v4i *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(v4i *p, v4i v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT>
__global__ __launch_bounds__(256) void kA(const v4i *__restrict__ src, v4i *__restrict__ dst, long long nv)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < nv) st<NT>(dst + i, ld<NT>(src + i));
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void kB(const v4i *__restrict__ src, v4i *__restrict__ dst, long long nv)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += U * stride) {
        v4i v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < nv) v[u] = ld<NT>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < nv) st<NT>(dst + i + u * stride, v[u]);
    }
}

// blocked: a workgroup owns U*256 consecutive vectors, thread t the vectors t, t+256, ...
template <int U, bool NT>
__global__ __launch_bounds__(256) void kF(const v4i *__restrict__ src, v4i *__restrict__ dst, long long nv)
{
    const long long i0 = (long long)blockIdx.x * (256 * U) + threadIdx.x;
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * 256 < nv) v[u] = ld<NT>(src + i0 + u * 256);
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * 256 < nv) st<NT>(dst + i0 + u * 256, v[u]);
}

// blocked with a dependent prologue: two dependent 8-byte loads (a table entry, then an entry it points to)
// decide the (zero) displacement of the chunk -- the shape of a compaction that looks its rows up first
template <int U, bool NT>
__global__ __launch_bounds__(256) void kG(const v4i *__restrict__ src, v4i *__restrict__ dst, long long nv,
                                          const long long *__restrict__ tab)
{
    const long long i0 = (long long)blockIdx.x * (256 * U) + threadIdx.x;
    const long long a = tab[blockIdx.x];                // 0..n-1: a second index
    const long long d = tab[a + (threadIdx.x & 7)];     // dependent; the table holds values < 1
    const long long j0 = i0 + (d >> 40);                // always 0, unknown to the compiler
    v4i v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * 256 < nv) v[u] = ld<NT>(src + j0 + u * 256);
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * 256 < nv) st<NT>(dst + i0 + u * 256, v[u]);
}

// misaligned: dst[i] = src[i + s], s in 1..3, loads at dword alignment
template <int U>
__global__ __launch_bounds__(256) void kC(const int *__restrict__ src, v4i *__restrict__ dst, long long nv, int s)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += U * stride) {
        V4U v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < nv) v[u] = *reinterpret_cast<const V4U *>(src + 4 * (i + u * stride) + s);
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < nv) { v4i o = {v[u].x, v[u].y, v[u].z, v[u].w}; dst[i + u * stride] = o; }
    }
}

// misaligned through aligned loads: the 16 bytes at dword offset s of (a, b)
__device__ __forceinline__ v4i shifted(v4i a, v4i b, int s)
{
    v4i o;
    o.x = s == 0 ? a.x : s == 1 ? a.y : s == 2 ? a.z : a.w;
    o.y = s == 0 ? a.y : s == 1 ? a.z : s == 2 ? a.w : b.x;
    o.z = s == 0 ? a.z : s == 1 ? a.w : s == 2 ? b.x : b.y;
    o.w = s == 0 ? a.w : s == 1 ? b.x : s == 2 ? b.y : b.z;
    return o;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void kD(const v4i *__restrict__ src, v4i *__restrict__ dst, long long nv, int s)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += U * stride) {
        v4i a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < nv) { a[u] = ld<NT>(src + i + u * stride); b[u] = ld<NT>(src + i + u * stride + 1); }
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < nv) st<NT>(dst + i + u * stride, shifted(a[u], b[u], s));
    }
}

// as D, the second vector taken from the next lane (DPP wave_shl:1 is not on every GFX9 part: ds_bpermute-free
// form through __shfl_down); lane 63 loads it
template <int U>
__global__ __launch_bounds__(256) void kE(const v4i *__restrict__ src, v4i *__restrict__ dst, long long nv, int s)
{
    const long long stride = (long long)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += U * stride) {
        v4i a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            a[u] = (i + u * stride < nv) ? src[i + u * stride] : v4i{0, 0, 0, 0};
            if (lane == 63 && i + u * stride < nv) b[u] = src[i + u * stride + 1];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            v4i n;
            n.x = __shfl_down(a[u].x, 1, 64); n.y = __shfl_down(a[u].y, 1, 64); n.z = __shfl_down(a[u].z, 1, 64); n.w = 0;
            if (lane != 63) b[u] = n;
            if (i + u * stride < nv) dst[i + u * stride] = shifted(a[u], b[u], s);
        }
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    const long long n = 1336366080ll;             // multiple of 4
    const long long nv = n / 4;
    int *src = nullptr, *dst = nullptr;
    CHECK(hipMalloc(&src, (n + 64) * 4));
    CHECK(hipMalloc(&dst, n * 4));
    CHECK(hipMemset(src, 1, (n + 64) * 4));
    CHECK(hipMemset(dst, 0, n * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 5;
    auto report = [&](const char *name, float ms) {
        printf("%-58s %7.3f ms  %5.2f TB/s (read+write)\n", name, ms / reps, 2.0 * 4 * n / (ms / reps) / 1e9);
        fflush(stdout);
    };
#define TIME(name, launch) do { auto f_ = [&] { launch; }; f_(); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) f_(); \
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms_; CHECK(hipEventElapsedTime(&ms_, e0, e1)); CHECK(hipGetLastError()); report(name, ms_); } while (0)
    const v4i *s4 = reinterpret_cast<const v4i *>(src);
    v4i *d4 = reinterpret_cast<v4i *>(dst);
    const int gA = (int)((nv + 255) / 256);
    TIME("A  aligned, one vector per thread", hipLaunchKernelGGL(kA<false>, dim3(gA), dim3(256), 0, 0, s4, d4, nv));
    TIME("A  aligned, one vector per thread, non-temporal", hipLaunchKernelGGL(kA<true>, dim3(gA), dim3(256), 0, 0, s4, d4, nv));
    for (int wg : {2048, 4096, 8192, 16384}) {
        char nm[128];
        snprintf(nm, sizeof nm, "B  aligned, grid-stride U=4, %d workgroups", wg);
        TIME(nm, hipLaunchKernelGGL((kB<4, false>), dim3(wg), dim3(256), 0, 0, s4, d4, nv));
        snprintf(nm, sizeof nm, "B  aligned, grid-stride U=4, %d workgroups, non-temporal", wg);
        TIME(nm, hipLaunchKernelGGL((kB<4, true>), dim3(wg), dim3(256), 0, 0, s4, d4, nv));
    }
    {
        const int g2 = (int)((nv + 511) / 512), g4 = (int)((nv + 1023) / 1024), g8 = (int)((nv + 2047) / 2048);
        TIME("F  aligned, blocked U=2", hipLaunchKernelGGL((kF<2, false>), dim3(g2), dim3(256), 0, 0, s4, d4, nv));
        TIME("F  aligned, blocked U=4", hipLaunchKernelGGL((kF<4, false>), dim3(g4), dim3(256), 0, 0, s4, d4, nv));
        TIME("F  aligned, blocked U=4, non-temporal", hipLaunchKernelGGL((kF<4, true>), dim3(g4), dim3(256), 0, 0, s4, d4, nv));
        TIME("F  aligned, blocked U=8", hipLaunchKernelGGL((kF<8, false>), dim3(g8), dim3(256), 0, 0, s4, d4, nv));
        TIME("F  aligned, blocked U=8, non-temporal", hipLaunchKernelGGL((kF<8, true>), dim3(g8), dim3(256), 0, 0, s4, d4, nv));
        long long *tab = nullptr;
        CHECK(hipMalloc(&tab, (size_t)(g2 + 16) * 8));
        CHECK(hipMemset(tab, 0, (size_t)(g2 + 16) * 8));
        TIME("G  blocked U=2 behind two dependent loads", hipLaunchKernelGGL((kG<2, false>), dim3(g2), dim3(256), 0, 0, s4, d4, nv, tab));
        TIME("G  blocked U=4 behind two dependent loads", hipLaunchKernelGGL((kG<4, false>), dim3(g4), dim3(256), 0, 0, s4, d4, nv, tab));
        TIME("G  blocked U=4 behind two dependent loads, non-temporal", hipLaunchKernelGGL((kG<4, true>), dim3(g4), dim3(256), 0, 0, s4, d4, nv, tab));
        TIME("G  blocked U=8 behind two dependent loads", hipLaunchKernelGGL((kG<8, false>), dim3(g8), dim3(256), 0, 0, s4, d4, nv, tab));
        TIME("G  blocked U=8 behind two dependent loads, non-temporal", hipLaunchKernelGGL((kG<8, true>), dim3(g8), dim3(256), 0, 0, s4, d4, nv, tab));
    }
    TIME("B  aligned, grid-stride U=8, 4096 workgroups", hipLaunchKernelGGL((kB<8, false>), dim3(4096), dim3(256), 0, 0, s4, d4, nv));
    for (int s = 1; s <= 3; s += 2) {
        char nm[128];
        snprintf(nm, sizeof nm, "C  source +%d dword, 4-byte aligned 16-B loads, U=4, 8192 wg", s);
        TIME(nm, hipLaunchKernelGGL(kC<4>, dim3(8192), dim3(256), 0, 0, src, d4, nv, s));
        snprintf(nm, sizeof nm, "D  source +%d dword, two aligned loads + select, U=4, 8192 wg", s);
        TIME(nm, hipLaunchKernelGGL((kD<4, false>), dim3(8192), dim3(256), 0, 0, s4, d4, nv, s));
        snprintf(nm, sizeof nm, "D  the same, non-temporal", s);
        TIME(nm, hipLaunchKernelGGL((kD<4, true>), dim3(8192), dim3(256), 0, 0, s4, d4, nv, s));
        snprintf(nm, sizeof nm, "E  source +%d dword, one aligned load + next lane's, U=4, 8192 wg", s);
        TIME(nm, hipLaunchKernelGGL(kE<4>, dim3(8192), dim3(256), 0, 0, s4, d4, nv, s));
    }
    TIME("C  source +1 dword, U=2, 16384 wg", hipLaunchKernelGGL(kC<2>, dim3(16384), dim3(256), 0, 0, src, d4, nv, 1));
    TIME("D  source +1 dword, U=2, 16384 wg", hipLaunchKernelGGL((kD<2, false>), dim3(16384), dim3(256), 0, 0, s4, d4, nv, 1));
    return 0;
}
