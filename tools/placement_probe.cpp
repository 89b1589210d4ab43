// Developer probe: where does the hardware put the workgroups of a launch?  Each block records the XCC / SE / CU it
// runs on and when it started and finished (wall_clock64), for a grid shaped like conv_igemm's 64x64 launches
// (256 threads, 16 KB LDS).  Prints blocks per CU (min / max / histogram) and how many blocks started late.
//   hipcc --offload-arch=gfx950 -O3 -x hip tools/placement_probe.cpp -o tools/placement_probe && tools/placement_probe 1568
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256) probe(unsigned* out, long long* t0, long long* t1, int spin) {
    __shared__ float lds[4096];
    const long long start = wall_clock64();
    unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // HW_REG_XCC_ID
    float acc = threadIdx.x;
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + lds[(threadIdx.x + i) & 4095];
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        t0[blockIdx.x] = start; t1[blockIdx.x] = wall_clock64();
    }
    if (acc == 12345.f) out[0] = 0;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 1568, spin = argc > 2 ? atoi(argv[2]) : 20000;
    unsigned* out; long long *t0, *t1;
    hipMalloc(&out, grid * 8); hipMalloc(&t0, grid * 8); hipMalloc(&t1, grid * 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, out, t0, t1, spin); hipDeviceSynchronize(); }
    std::vector<unsigned> h(2 * grid); std::vector<long long> a(grid), b(grid);
    hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
    hipMemcpy(a.data(), t0, grid * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), t1, grid * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu; long long first = a[0], last = b[0];
    for (int i = 0; i < grid; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        if (a[i] < first) first = a[i];
        if (b[i] > last) last = b[i];
    }
    std::map<int, int> hist; int mn = 1 << 30, mx = 0;
    for (auto& kv : per_cu) { hist[kv.second]++; mn = kv.second < mn ? kv.second : mn; mx = kv.second > mx ? kv.second : mx; }
    int late = 0; long long dur = 0;
    for (int i = 0; i < grid; ++i) { dur += b[i] - a[i]; if (a[i] - first > (last - first) / 4) late++; }
    printf("grid %d: %zu distinct CUs, blocks per CU min %d max %d; histogram:", grid, per_cu.size(), mn, mx);
    for (auto& kv : hist) printf(" %dx%d", kv.first, kv.second);
    printf("\n  span %.1f us (100 MHz clock), mean block %.1f us, blocks starting after 25%% of the span: %d\n",
           (last - first) / 100.0, dur / 100.0 / grid, late);
    return 0;
}
