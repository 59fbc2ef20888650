// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this library uses
// (MI355X_MICROARCH.md §HBM: FETCH_SIZE halves 16-B/lane streaming reads; other widths are to be
// calibrated on a known byte count).  Each kernel streams a 1 GiB buffer once with one width:
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/micro/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v2 __attribute__((ext_vector_type(2)));
typedef int v4 __attribute__((ext_vector_type(4)));
template <typename T> __global__ void k_read(const T *p, size_t n, int *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    int acc = 0;
    for (; i < n; i += stride) { T v = p[i]; const int *q = (const int *)&v; for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= q[k]; }
    if (acc == 0x12345678) *sink = acc;
}
// random 64-byte rows of dwords (the B-row gather of the wave kernels: 16 consecutive dwords per row,
// one row per quarter wave), every row read exactly once
__global__ void k_gather64(const int *p, const int *perm, size_t rows, int *sink)
{
    size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, stride = ((size_t)gridDim.x * blockDim.x) >> 4;
    int acc = 0;
    for (; g < rows; g += stride) acc ^= p[(size_t)perm[g] * 16 + (threadIdx.x & 15)];
    if (acc == 0x12345678) *sink = acc;
}
template <typename T> __global__ void k_write(T *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    T v; int *q = (int *)&v; for (unsigned k = 0; k < sizeof(T) / 4; k++) q[k] = (int)i;
    for (; i < n; i += stride) p[i] = v;
}
int main()
{
    const size_t bytes = 1ull << 30;
    void *buf; int *sink, *perm;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
    const size_t rows = bytes / 64;
    hipMalloc(&perm, rows * 4);
    int *h = (int *)malloc(rows * 4);
    for (size_t i = 0; i < rows; i++) h[i] = (int)i;
    unsigned long long s = 88172645463325252ull;
    for (size_t i = rows - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; size_t j = s % (i + 1); int t = h[i]; h[i] = h[j]; h[j] = t; }
    hipMemcpy(perm, h, rows * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        k_read<int><<<4096, 256>>>((const int *)buf, bytes / 4, sink);
        k_read<v2><<<4096, 256>>>((const v2 *)buf, bytes / 8, sink);
        k_read<v4><<<4096, 256>>>((const v4 *)buf, bytes / 16, sink);
        k_gather64<<<4096, 256>>>((const int *)buf, perm, rows, sink);
        k_write<int><<<4096, 256>>>((int *)buf, bytes / 4);
        k_write<v4><<<4096, 256>>>((v4 *)buf, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes (k_gather64 additionally reads %zu bytes of row indices)\n", bytes, rows * 4);
    return 0;
}
