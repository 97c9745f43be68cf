// cu_mask.hip — which CUs does a stream created with hipExtStreamCreateWithCUMask use?  (round 6: de_tuning.v6_cu_withhold clears the TOP 8 x n bits of the mask and
// assumes that this withholds n CUs of every XCD.)  Launches workgroups that record (XCC_ID, SE, SH, CU) on a plain stream and on masked streams and prints, per XCD,
// the number of distinct CUs seen.   hipcc --offload-arch=gfx950 -O2 tools/micro/cu_mask.hip -o tools/micro/cu_mask && tools/micro/cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

__global__ void where(uint32_t* out) {
    if (threadIdx.x == 0) {
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x] = ((xcc & 15u) << 16) | ((hw >> 8) & 0xffu);      // CU_ID 11:8, SH_ID 12, SE_ID 15:13
    }
    // keep the workgroup resident for a while so that the launch spreads over every CU it may use
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 20000ull) __builtin_amdgcn_s_sleep(8);
}

static void run(const char* name, hipStream_t s, uint32_t* d, int n) {
    std::vector<uint32_t> h(n);
    hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::set<uint32_t> per[16];
    for (uint32_t v : h) per[(v >> 16) & 15].insert(v & 0xffu);
    printf("%-28s", name);
    int total = 0;
    for (int x = 0; x < 8; ++x) { printf(" xcd%d:%2zu", x, per[x].size()); total += (int)per[x].size(); }
    printf("  total %d\n", total);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cus = p.multiProcessorCount, n = 8192;
    uint32_t* d; hipMalloc(&d, n * 4);
    hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    run("plain stream", s0, d, n);
    for (int withhold : {1, 2, 4}) {
        uint32_t mask[16]; memset(mask, 0, sizeof(mask));
        const int keep = n_cus - 8 * withhold;
        for (int i = 0; i < keep; ++i) mask[i >> 5] |= 1u << (i & 31);
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((n_cus + 31) / 32), mask);
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); continue; }
        char name[64]; snprintf(name, sizeof(name), "top %d bits cleared", 8 * withhold);
        run(name, s, d, n);
        hipStreamDestroy(s);
    }
    {   // the other guess: bits of one XCD are contiguous (32 per XCD): clear the last bit of each 32-bit word
        uint32_t mask[16]; memset(mask, 0xff, sizeof(mask));
        for (int w = 0; w < (n_cus + 31) / 32; ++w) mask[w] &= 0x7fffffffu;
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)((n_cus + 31) / 32), mask) == hipSuccess) { run("bit 31 of every word cleared", s, d, n); hipStreamDestroy(s); }
    }
    return 0;
}
