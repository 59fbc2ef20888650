// FETCH_SIZE calibration on the numeric kernel's ACTUAL access: rows of 16..20 consecutive dwords
// read by one lane per dword, rows at 64-byte-aligned / 32-byte-aligned / random dword-aligned starts, every row read
// once, rows spread over a 1 GiB buffer (25 % coverage, so that neighbouring rows rarely share a line).  The
// algorithmic bytes of each kernel are rows * L * 4 (+ 8 bytes per row of start offsets, streamed).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib2 tools/micro/fetch_calib2.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int L>   // lanes (= dwords) per row; 64 / L rows per wave
__global__ void k_rows(const int *__restrict__ p, const long long *__restrict__ starts, size_t rows, int *sink)
{
    constexpr int RPW = 64 / L;
    const int lane = threadIdx.x & 63;
    const int sub = lane / L, k = lane % L;
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    int acc = 0;
    for (size_t r = wave * RPW + sub; r < rows; r += nwaves * RPW)
        if (sub < RPW) acc ^= p[starts[r] + k];
    if (acc == 0x12345678) *sink = acc;
}

static unsigned long long s_rng = 88172645463325252ull;
static unsigned long long rnd() { s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17; return s_rng; }

int main()
{
    const size_t bytes = 1ull << 30, dwords = bytes / 4;
    const size_t rows = bytes / 64 / 4;                       // 25 % coverage
    int *buf, *sink;
    long long *d_starts[4];
    hipMalloc(&buf, bytes + 256); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes + 256);
    long long *h = (long long *)malloc(rows * sizeof(long long));
    const char *names[4] = {"64B-aligned", "32B-aligned", "16B-aligned", "dword-aligned (random)"};
    const int gran[4] = {16, 8, 4, 1};
    for (int v = 0; v < 4; v++) {
        for (size_t i = 0; i < rows; i++) h[i] = (long long)((rnd() % ((dwords - 64) / gran[v])) * gran[v]);
        hipMalloc(&d_starts[v], rows * sizeof(long long));
        hipMemcpy(d_starts[v], h, rows * sizeof(long long), hipMemcpyHostToDevice);
    }
    for (int rep = 0; rep < 2; rep++)
        for (int v = 0; v < 4; v++) {
            // one launch per (alignment, row length): the kernel name carries L, the dispatch order the alignment
            k_rows<16><<<8192, 256>>>(buf, d_starts[v], rows, sink);
            k_rows<20><<<8192, 256>>>(buf, d_starts[v], rows * 3 / 4, sink);
        }
    hipDeviceSynchronize();
    printf("dispatch order per repetition: for alignment in [%s, %s, %s, %s]: k_rows<16> (%zu rows, %zu bytes), k_rows<20> (%zu rows, %zu bytes); "
           "each also streams 8 bytes of start offset per row\n", names[0], names[1], names[2], names[3],
           rows, rows * 64, rows * 3 / 4, rows * 3 / 4 * 80);
    return 0;
}
