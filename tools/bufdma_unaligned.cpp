// Developer probe: `buffer_load_dwordx4 ... lds` from a global address that is only 4-byte aligned (voffset = 16*lane + 4*shift),
// and with a voffset that wraps "below" the buffer start (negative shift on lane 0) -- the two things the shifted-row im2col
// staging of conv_igemm relies on.  Expected: LDS float i == src[i + shift] (0 where out of range).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* src, float* out, int nrec_bytes, int shift) {
    __shared__ __attribute__((aligned(16))) float buf[256];
    for (int i = threadIdx.x; i < 256; i += 64) buf[i] = 7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec_bytes, 0x00020000);
    const unsigned voff = (unsigned)((int)threadIdx.x * 16 + shift * 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)&buf[0], 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = buf[i];
}
int main() {
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = 100.f + i;
    float *d, *o; hipMalloc(&d, sizeof h); hipMalloc(&o, 1024); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int shift : {0, 1, 2, 3, -1, -15, 13}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d + 64, o, 300 * 4, shift);       // base = &h[64], 300 records
        float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) {
            const int s = i + shift;
            const float want = (s >= 0 && s < 300) ? 100.f + 64 + s : 0.f;
            if (r[i] != want) { if (bad < 6) printf("  shift %d: lds[%d] = %.0f want %.0f\n", shift, i, r[i], want); ++bad; }
        }
        printf("shift %3d: %s (%d mismatches)  first floats: %.0f %.0f %.0f %.0f %.0f\n", shift, bad ? "MISMATCH" : "ok", bad, r[0], r[1], r[2], r[3], r[4]);
    }
    return 0;
}
