// lds_ops.hip -- what ONE LDS wave-instruction costs the LDS pipe of a CU, by kind and address pattern (gfx950).
// Every kernel issues ITERS x UNROLL instructions of one kind per wave, 20 waves per CU (5 workgroups of 4 waves, each wave on
// its own 8 KiB slice), addresses from a per-lane LCG (random) or a fixed lane stride (blocked); the figure printed is
// LDS-pipe cycles per wave-instruction = kernel time x clock x CUs / (waves x instructions), i.e. the reciprocal throughput
// of the CU's LDS under the occupancy the accumulate kernels run at.
//   hipcc --offload-arch=gfx950 -O3 -o lds_ops tools/micro/lds_ops.hip && ./lds_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32;
typedef u32 v4 __attribute__((ext_vector_type(4)));
constexpr int WORDS = 2048;      // 8 KiB per wave
constexpr int ITERS = 2000, UNROLL = 8;

template <int KIND, int STRIDE>
__global__ __launch_bounds__(256) void k(u32 *sink)
{
    __shared__ __attribute__((aligned(16))) u32 s[4][WORDS];
    u32 *p = s[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < WORDS; i += 64) p[i] = i;
    __syncthreads();
    // eight random words per lane, drawn once; per iteration they move by an odd step (two cheap VALU instructions per
    // access: the first version of this test drew a fresh LCG value per access and measured v_mul_lo_u32, not the LDS)
    u32 ra[UNROLL], acc = 0;
    {
        u32 a = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { a = a * 1664525u + 1013904223u; ra[u] = a >> 12; }
    }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const u32 r = (ra[u] + it * 67) & (WORDS - 1);               // random word
            const u32 b = ((lane * STRIDE + u * 4 + it * 8) & (WORDS - 4));   // blocked: lane stride STRIDE words
            const u32 a = r ^ it;
            if (KIND == 0) acc += p[r];                                  // ds_read_b32 random
            if (KIND == 1) p[r] = a;                                     // ds_write_b32 random
            if (KIND == 2) atomicOr(&p[r], a);                           // ds_or_b32 (no return) random
            if (KIND == 3) acc += atomicAdd(&p[r], 1u);                  // ds_add_rtn_u32 random
            if (KIND == 4) acc += p[(lane + u * 64 + it) & (WORDS - 1)]; // ds_read_b32 coalesced
            if (KIND == 5) { const v4 v = *reinterpret_cast<const v4 *>(p + b); acc += v.x + v.w; }   // ds_read_b128 blocked
            if (KIND == 6) { const v4 v = {a, a, a, a}; *reinterpret_cast<v4 *>(p + b) = v; }          // ds_write_b128 blocked
            if (KIND == 7) acc += p[r & 31];                             // ds_read_b32, 32 distinct words (few addresses: broadcast)
            if (KIND == 8) { acc += p[r]; acc += ((unsigned short *)p)[r * 2 + 1]; }   // b32 + u16 at the same index (rank + prefix)
            if (KIND == 9) acc += p[(r & ~63u) | (lane & 63)];           // random row, own bank: conflict-free "random"
            if (KIND == 10) { const unsigned long long v = *reinterpret_cast<const unsigned long long *>(p + (r & ~1u)); acc += (u32)v + (u32)(v >> 32); }   // ds_read_b64 random
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int KIND, int STRIDE> static void run(const char *name, int per_iter)
{
    u32 *sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 5;
    k<KIND, STRIDE><<<grid, 256>>>(sink);
    hipEventRecord(e0);
    k<KIND, STRIDE><<<grid, 256>>>(sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_cu = 20.0 * ITERS * UNROLL * per_iter;        // 20 waves per CU
    printf("%-58s %8.3f ms  -> %5.2f cycles per wave-instruction per CU (2.4 GHz assumed)\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_cu);
    hipFree(sink);
}

int main()
{
    run<4, 0>("ds_read_b32 coalesced", 1);
    run<7, 0>("ds_read_b32 few addresses (broadcast)", 1);
    run<9, 0>("ds_read_b32 random row, lane's own bank (no conflicts)", 1);
    run<0, 0>("ds_read_b32 random", 1);
    run<8, 0>("ds_read_b32 + ds_read_u16 random, same index", 2);
    run<10, 0>("ds_read_b64 random (8-byte aligned)", 1);
    run<1, 0>("ds_write_b32 random", 1);
    run<2, 0>("ds_or_b32 random (no return)", 1);
    run<3, 0>("ds_add_rtn_u32 random", 1);
    run<5, 4>("ds_read_b128 blocked, lane stride 16 B", 1);
    run<5, 8>("ds_read_b128 blocked, lane stride 32 B", 1);
    run<5, 12>("ds_read_b128 blocked, lane stride 48 B", 1);
    run<5, 16>("ds_read_b128 blocked, lane stride 64 B", 1);
    run<5, 32>("ds_read_b128 blocked, lane stride 128 B", 1);
    run<6, 4>("ds_write_b128 blocked, lane stride 16 B", 1);
    run<6, 8>("ds_write_b128 blocked, lane stride 32 B", 1);
    run<6, 12>("ds_write_b128 blocked, lane stride 48 B", 1);
    run<6, 16>("ds_write_b128 blocked, lane stride 64 B", 1);
    run<6, 32>("ds_write_b128 blocked, lane stride 128 B", 1);
    return 0;
}
