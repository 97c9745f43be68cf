// gfx950: what does rocprofv3's FETCH_SIZE count for THIS kernel's access pattern?  (MI355X_MICROARCH.md §HBM:
// "calibrate on a known byte count in your own access pattern".)
//
// Three kernels over a 2 GiB buffer (8x the 256 MiB Infinity Cache), every byte touched at most once per launch:
//   stream   16 B per lane, coalesced             -> known traffic = buffer bytes
//   sparse   ONE dword per 128-byte line, lines in a hashed (scattered) order: the render kernel's tap pattern with a cold
//            cache -> known footprint = one line per load; whether a miss moves 32, 64 or 128 bytes is what is measured
//   sparse64 one dword per 64-byte half line, both halves of every line touched by different wavefronts
// The program also times `sparse` at full occupancy: lines per second x 128 B above the 8 TB/s peak would prove that a
// dword miss does not move a whole 128-byte line.
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_calib     (tools/fetch_calib.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }

__global__ void __launch_bounds__(256) stream(const uint4* __restrict__ buf, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// line index = bijection of the load index (multiplication by an odd constant modulo 2^k, then xor-shift): every line once
template <int STRIDE_BYTES>
__global__ void __launch_bounds__(256) sparse(const uint32_t* __restrict__ buf, uint32_t n_lines_log2, uint32_t* sink) {
    const uint32_t mask = (1u << n_lines_log2) - 1u;
    uint32_t acc = 0;
    const uint32_t n = 1u << n_lines_log2;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t j = (i * 0x9E3779B1u) & mask;
        j ^= j >> (n_lines_log2 / 2);                   // xor-shift by half the width: a bijection on k-bit words
        j = (j * 0x85EBCA6Bu) & mask;
        acc ^= buf[(size_t)j * (STRIDE_BYTES / 4) + (mix(i) & (STRIDE_BYTES / 4 - 1))];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    uint32_t* buf; uint32_t* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 64);
    hipMemset(buf, 1, bytes); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const int blocks = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); stream<<<blocks, 256>>>((const uint4*)buf, bytes / 16, sink); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("stream   : %.3f ms, known bytes %zu -> %.0f GB/s\n", ms, bytes, bytes / (ms * 1e-3) / 1e9);
        hipEventRecord(e0); sparse<128><<<blocks, 256>>>(buf, 24, sink); hipEventRecord(e1); hipDeviceSynchronize();   // 2^24 lines of 128 B = 2 GiB
        hipEventElapsedTime(&ms, e0, e1);
        printf("sparse128: %.3f ms, %u loads, one per 128-B line -> %.2f G lines/s (x128 B = %.0f GB/s, x64 B = %.0f GB/s)\n", ms, 1u << 24,
               (double)(1u << 24) / (ms * 1e-3) / 1e9, (double)(1u << 24) * 128 / (ms * 1e-3) / 1e9, (double)(1u << 24) * 64 / (ms * 1e-3) / 1e9);
        hipEventRecord(e0); sparse<64><<<blocks, 256>>>(buf, 25, sink); hipEventRecord(e1); hipDeviceSynchronize();    // 2^25 half lines of 64 B = 2 GiB
        hipEventElapsedTime(&ms, e0, e1);
        printf("sparse64 : %.3f ms, %u loads, one per 64-B half line -> %.2f G loads/s (x64 B = %.0f GB/s)\n", ms, 1u << 25,
               (double)(1u << 25) / (ms * 1e-3) / 1e9, (double)(1u << 25) * 64 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
