// op_rates.hip — issue cost of the integer / special-function instructions in render_kernel_v6's step bodies relative to v_fma_f32, at 6 waves per SIMD with
// 4 independent chains per wave (round 6): are v_mul_lo_u32 (the generator's scrambler), v_mad_u64_u32 (the tap's tile index), v_rcp / v_rsq (normalisation)
// quarter-rate on gfx950, and would v_mad_u32_u24 for the tile index be cheaper?   hipcc --offload-arch=gfx950 -O2 tools/micro/op_rates.hip -o tools/micro/op_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(NAME, ASM)                                                                                   \
    __global__ void NAME(uint32_t* out, int iters) {                                                      \
        uint32_t a0 = threadIdx.x + 1u, a1 = threadIdx.x + 3u, a2 = threadIdx.x + 5u, a3 = threadIdx.x + 7u, b = 0x9E3779BBu; \
        for (int i = 0; i < iters; ++i) {                                                                 \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                               \
                asm volatile(ASM : "+v"(a0) : "v"(b));                                                    \
                asm volatile(ASM : "+v"(a1) : "v"(b));                                                    \
                asm volatile(ASM : "+v"(a2) : "v"(b));                                                    \
                asm volatile(ASM : "+v"(a3) : "v"(b));                                                    \
            }                                                                                             \
        }                                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;                                   \
    }
BODY(k_fma, "v_fma_f32 %0, %0, %1, %0")
BODY(k_add, "v_add_u32 %0, %0, %1")
BODY(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
BODY(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
BODY(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %0")
BODY(k_rcp, "v_rcp_f32 %0, %0")
BODY(k_rsq, "v_rsq_f32 %0, %0")
BODY(k_cvt, "v_cvt_f32_u32 %0, %0")
BODY(k_alignbit, "v_alignbit_b32 %0, %0, %0, 6")
BODY(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
BODY(k_floor, "v_floor_f32 %0, %0")
__global__ void k_mad64(uint32_t* out, int iters) {
    unsigned long long a0 = threadIdx.x + 1u, a1 = threadIdx.x + 3u, a2 = threadIdx.x + 5u, a3 = threadIdx.x + 7u; uint32_t b = 2700u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"((uint32_t)threadIdx.x), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"((uint32_t)threadIdx.x), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"((uint32_t)threadIdx.x), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"((uint32_t)threadIdx.x), "v"(b) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3);
}

template <typename K>
static double run(const char* name, K kern, uint32_t* out, double base) {
    const int waves = 6, blocks = 256 * 4 * waves, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 64>>>(out, 100); hipDeviceSynchronize();
    hipEventRecord(e0); kern<<<blocks, 64>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 32 * waves);
    printf("%-16s %.2f ns per instruction per SIMD at 6 waves%s\n", name, ns, base > 0 ? "" : "  (the unit)");
    if (base > 0) printf("%-16s = %.2f x v_fma_f32\n", "", ns / base);
    return ns;
}
int main() {
    uint32_t* out; hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    const double f = run("v_fma_f32", k_fma, out, 0);
    run("v_add_u32", k_add, out, f); run("v_mul_lo_u32", k_mul_lo, out, f); run("v_mul_u32_u24", k_mul_u24, out, f); run("v_mad_u32_u24", k_mad_u24, out, f);
    run("v_mad_u64_u32", k_mad64, out, f); run("v_rcp_f32", k_rcp, out, f); run("v_rsq_f32", k_rsq, out, f); run("v_cvt_f32_u32", k_cvt, out, f);
    run("v_alignbit_b32", k_alignbit, out, f); run("v_cndmask_b32", k_cndmask, out, f); run("v_floor_f32", k_floor, out, f);
    return 0;
}
