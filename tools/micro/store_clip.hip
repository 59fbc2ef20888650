// store_clip.hip -- does a raw-buffer dwordx3 store clip per DWORD at num_records (gfx950)?
// Each lane stores {0x11,0x22,0x33}+lane at byte offset 12*lane through a descriptor of `bytes` bytes; the host prints
// which dwords of a guard-filled buffer changed.  build: hipcc --offload-arch=gfx950 -O3 -o store_clip store_clip.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v3i __attribute__((ext_vector_type(3)));
__global__ void k(int *out, int bytes)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)out, (short)0, bytes, 0x00020000);
    const int lane = threadIdx.x;
    v3i v = {0x1100 + lane, 0x2200 + lane, 0x3300 + lane};
    __builtin_amdgcn_raw_buffer_store_b96(v, rs, lane * 12, 0, 0);
}
int main()
{
    int *d; hipMalloc(&d, 4096);
    for (int bytes : {0, 4, 8, 12, 16, 20, 28, 40}) {
        hipMemset(d, 0xff, 4096);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, bytes);
        int h[32]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("num_records %2d bytes: dwords written:", bytes);
        for (int i = 0; i < 32; i++) if (h[i] != -1) printf(" %d", i);
        printf("\n");
    }
    return 0;
}
