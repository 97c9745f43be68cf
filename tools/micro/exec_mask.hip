// Does a wave64 VALU instruction get cheaper when whole 16-lane quarters of EXEC are off?  (gfx950 micro-benchmark)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(float* out, unsigned long long mask, int iters, unsigned long long* cyc) {
    int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-3f, b = 0.999f, c = 1e-4f;
    float a2 = a + 1.f, a3 = a + 2.f, a4 = a + 3.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a = __builtin_fmaf(a, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c); a4 = __builtin_fmaf(a4, b, c);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    int blocks = 256 * 4, threads = 256;   // 4 waves per SIMD, every SIMD busy
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, 8);
    unsigned long long masks[32]; const char* names[32]; char buf[32][32]; int nm = 0;
    int ns[] = {1, 4, 8, 9, 10, 12, 14, 15, 16, 17, 24};
    for (int n : ns) { masks[nm] = ((1ull << n) - 1); snprintf(buf[nm], 32, "first %d lanes", n); names[nm] = buf[nm]; ++nm; }
    masks[nm] = 0x000F000F000F000Full >> 0 & 0x0003000300030003ull; names[nm++] = "2 lanes per quarter (8)";
    masks[nm] = 0x0007000700070007ull; names[nm++] = "3 lanes per quarter (12)";
    masks[nm] = 0x00FF0000000000FFull; names[nm++] = "8 in q0 + 8 in q3";
    masks[nm] = 0x8000000000000000ull | 0xFFull; names[nm++] = "8 in q0 + lane 63";
    for (int m = 0; m < nm; ++m) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<blocks, threads>>>(out, masks[m], 100, cyc); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, threads>>>(out, masks[m], 20000, cyc); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr = 20000.0 * 64;   // VALU instructions per wave
        printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (4 waves/SIMD)\n", names[m], ms, ms * 1e6 / (instr * 4));
    }
    return 0;
}
