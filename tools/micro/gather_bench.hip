// gather_bench.hip -- how fast can the prepass turn A.col_idx into B-row extents?
// Times, for 67M uniformly random column indices into a 4.2M-row B:
//   A  8-byte pair gather from the int row_ptr table (16.8 MB)             [what k_row_work does]
//   B  1-byte gather from a u8 degree table (4.2 MB)                        [enough for F_i]
//   C  4-byte gather from a coarse row_ptr (every 64 rows, 262 KB) + the 64-byte line of u8
//      degrees of the group, summed below the row                          [start AND length]
// build: hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct __attribute__((packed, aligned(4))) Int2U { int x, y; };

__global__ void kA(const int *__restrict__ col, const int *__restrict__ rp, int2 *__restrict__ out, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        int j[4];
        Int2U p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) j[u] = col[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = *reinterpret_cast<const Int2U *>(rp + j[u]);
#pragma unroll
        for (int u = 0; u < 4; u++) out[i + u * stride] = make_int2(p[u].x, p[u].y - p[u].x);
    }
}

__global__ void kB(const int *__restrict__ col, const unsigned char *__restrict__ deg, int *__restrict__ out, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        int j[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) j[u] = col[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) d[u] = deg[j[u]];
#pragma unroll
        for (int u = 0; u < 4; u++) out[i + u * stride] = d[u];
    }
}

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned sum_below(v4u q, int k)   // sum of bytes 0..k-1 of the 16 bytes in q
{
    unsigned s = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        int nb = k - 4 * d;
        nb = nb < 0 ? 0 : (nb > 4 ? 4 : nb);
        const unsigned m = nb == 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
        s = __builtin_amdgcn_sad_u8(q[d] & m, 0u, s);
    }
    return s;
}

__global__ void kC(const int *__restrict__ col, const int *__restrict__ rp64, const unsigned char *__restrict__ deg,
                   int2 *__restrict__ out, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i + 1 * stride < n; i += 2 * stride) {
        int j[2], base[2];
        v4u q[2][4];
#pragma unroll
        for (int u = 0; u < 2; u++) j[u] = col[i + u * stride];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            base[u] = rp64[j[u] >> 6];
            const v4u *line = reinterpret_cast<const v4u *>(deg + ((long long)(j[u] >> 6) << 6));
#pragma unroll
            for (int t = 0; t < 4; t++) q[u][t] = line[t];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = j[u] & 63;
            unsigned s = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) s += sum_below(q[u][t], k - 16 * t);
            const unsigned len = (reinterpret_cast<const unsigned char *>(&q[u][0]))[k];
            out[i + u * stride] = make_int2(base[u] + (int)s, (int)len);
        }
    }
}

// C16: coarse row_ptr every 16 rows (1 MB) + ONE 16-byte load of the group's u8 degrees
__global__ void kD(const int *__restrict__ col, const int *__restrict__ rp16, const unsigned char *__restrict__ deg,
                   int2 *__restrict__ out, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        int j[4], base[4];
        v4u q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) j[u] = col[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            base[u] = rp16[j[u] >> 4];
            q[u] = *reinterpret_cast<const v4u *>(deg + ((long long)(j[u] >> 4) << 4));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = j[u] & 15;
            const unsigned s = sum_below(q[u], k);
            const unsigned len = (q[u][k >> 2] >> (8 * (k & 3))) & 255u;
            out[i + u * stride] = make_int2(base[u] + (int)s, (int)len);
        }
    }
}

int main()
{
    const int nrows = 1 << 22;
    const long long nnz = 67108864ll;
    std::vector<int> col(nnz), rp(nrows + 1), rp64(nrows / 64 + 1);
    std::vector<unsigned char> deg(nrows);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (long long i = 0; i < nnz; i++) col[i] = (int)(rnd() % nrows);
    rp[0] = 0;
    for (int r = 0; r < nrows; r++) { deg[r] = (unsigned char)(8 + rnd() % 17); rp[r + 1] = rp[r] + deg[r]; }
    for (int g = 0; g <= nrows / 64; g++) rp64[g] = rp[g * 64 < nrows ? g * 64 : nrows];
    std::vector<int> rp16(nrows / 16 + 1);
    for (int g = 0; g <= nrows / 16; g++) rp16[g] = rp[g * 16 < nrows ? g * 16 : nrows];
    int *d_rp16; hipMalloc(&d_rp16, rp16.size() * 4); hipMemcpy(d_rp16, rp16.data(), rp16.size() * 4, hipMemcpyHostToDevice);
    int *d_col, *d_rp, *d_rp64, *d_out1;
    unsigned char *d_deg;
    int2 *d_out2;
    hipMalloc(&d_col, nnz * 4); hipMalloc(&d_rp, (nrows + 1) * 4); hipMalloc(&d_rp64, rp64.size() * 4);
    hipMalloc(&d_deg, nrows + 64); hipMalloc(&d_out1, nnz * 4); hipMalloc(&d_out2, nnz * 8);
    hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_rp, rp.data(), (nrows + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_rp64, rp64.data(), rp64.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_deg, deg.data(), nrows, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 16, block = 256;
    for (int which = 0; which < 4; which++) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(kA, dim3(grid), dim3(block), 0, 0, d_col, d_rp, d_out2, nnz);
            if (which == 1) hipLaunchKernelGGL(kB, dim3(grid), dim3(block), 0, 0, d_col, d_deg, d_out1, nnz);
            if (which == 3) hipLaunchKernelGGL(kD, dim3(grid), dim3(block), 0, 0, d_col, d_rp16, d_deg, d_out2, nnz);
            if (which == 2) hipLaunchKernelGGL(kC, dim3(grid), dim3(block), 0, 0, d_col, d_rp64, d_deg, d_out2, nnz);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%s  %.3f ms  (%.1f G gathers/s)\n", which == 0 ? "A pair/int-table " : which == 1 ? "B u8 degree      " : which == 2 ? "C rp64 + u8 line " : "D rp16 + 16 B    ",
               best, nnz / best / 1e6);
    }
    // check C against A on a sample
    std::vector<int2> oa(1 << 16), oc(1 << 16);
    hipLaunchKernelGGL(kA, dim3(grid), dim3(block), 0, 0, d_col, d_rp, d_out2, nnz);
    hipMemcpy(oa.data(), d_out2, oa.size() * 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(kD, dim3(grid), dim3(block), 0, 0, d_col, d_rp16, d_deg, d_out2, nnz);
    hipMemcpy(oc.data(), d_out2, oc.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < oa.size(); i++) bad += (oa[i].x != oc[i].x || oa[i].y != oc[i].y);
    printf("D vs A mismatches in sample: %d\n", bad);
    return 0;
}
