// Developer probe: does an out-of-range lane of `buffer_load_dword ... lds` write 0 to LDS or leave it?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* src, float* out, int nrec_bytes, int use_soffset) {
    __shared__ float buf[256];
    buf[threadIdx.x] = 7.f; buf[threadIdx.x + 64] = 7.f; buf[threadIdx.x + 128] = 7.f; buf[threadIdx.x + 192] = 7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec_bytes, 0x00020000);
    unsigned voff = threadIdx.x * 4;
    if (threadIdx.x % 3 == 0) voff = 0x80000000u;          // out of range lanes
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)&buf[0], 4, voff, use_soffset ? 64 * 4 : 0, 0, 0);
    // 16-byte variant into the second half
    unsigned voff4 = threadIdx.x * 16;
    if (threadIdx.x >= 8) voff4 = 0xFFFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)&buf[64], 16, voff4 & (threadIdx.x < 12 ? 0xFFFFFFFFu : 0xFFFFFFFFu), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = buf[i];
}
int main() {
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = 100.f + i;
    float *d, *o; hipMalloc(&d, sizeof h); hipMalloc(&o, 1024); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int so = 0; so < 2; ++so) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 512 * 4, so);
        float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        printf("soffset=%d dword DMA lanes 0..7:", so); for (int i = 0; i < 8; ++i) printf(" %.0f", r[i]); printf("\n");
        printf("   x4 DMA floats 64..71:"); for (int i = 64; i < 72; ++i) printf(" %.0f", r[i]);
        printf(" | 92..99:"); for (int i = 92; i < 100; ++i) printf(" %.0f", r[i]); printf("\n");
    }
    // range check with soffset pushing past the end
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 100 * 4, 1);
    float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    printf("nrec=100 floats, soffset=64: lanes 30..40:"); for (int i = 30; i < 41; ++i) printf(" %.0f", r[i]); printf("\n");
    return 0;
}
