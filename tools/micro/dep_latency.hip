// gfx950: issue interval of dependent vs independent wave64 f32 FMAs, at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(float* out, int iters) {
    float a[CHAINS];
    for (int c = 0; c < CHAINS; ++c) a[c] = 1.0f + threadIdx.x * 1e-3f + c;
    const float b = 0.999f, d = 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 32 / CHAINS; ++j)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fmaf(a[c], b, d);
    }
    float s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(float* out, int waves_per_simd) {
    // 256 CUs x 4 SIMDs; one 64-thread block per wave
    int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<blocks, 64>>>(out, 100); hipDeviceSynchronize();
    const int iters = 40000;
    hipEventRecord(e0); k<CHAINS><<<blocks, 64>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per_simd_instr = (double)iters * 32 * waves_per_simd;
    printf("chains %d waves/SIMD %d: %.2f ns per instruction per SIMD (%.2f ns between a wave's instructions)\n", CHAINS, waves_per_simd,
           ms * 1e6 / per_simd_instr, ms * 1e6 / ((double)iters * 32));
}
int main() {
    float* out; hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    for (int w = 1; w <= 8; ++w) { run<1>(out, w); run<4>(out, w); }
    return 0;
}
