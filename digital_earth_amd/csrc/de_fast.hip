// de_fast.hip — SECOND translation unit of libdigitalearth_hip.so: render_kernel_v6 compiled with -DDE_FAST_MATH semantics (de_math.h: the hardware's
// transcendental units and shorter polynomials instead of the arithmetic contract's fixed sequences), selected per context by DE_FLAG_FAST_MATH.
// Opt-in and NOT bit-reproducible against the CPU oracle (what bit-exactness costs, priced — profiles/r5_fast_math.md).
// Device code of the two units never meets (no relocatable device code); on the host side the kernel and its init kernel get their own names
// so that the two units' launch stubs stay apart.
#define DE_FAST_MATH 1
#define v6_kernel v6_kernel_fast_math
#define v6_init_kernel v6_init_kernel_fast_math
#include "de_stages.h"
#include "render_kernel_v6.hip"

// launched by de_launch.h: launch_v6 (the control words and RenderArgs are initialised there, by the contract unit's v6_init_kernel)
hipError_t de_fast_launch_v6(const bs::Args& A, const FrameConsts* fc, dim3 grid, dim3 block, hipStream_t stream, bool clamp, bool tail) {
    if (tail) {
        if (clamp) hipLaunchKernelGGL((v6_kernel<true, false, true>), grid, block, 0, stream, A, fc);
        else hipLaunchKernelGGL((v6_kernel<false, false, true>), grid, block, 0, stream, A, fc);
    } else {
        if (clamp) hipLaunchKernelGGL((v6_kernel<true, false, false>), grid, block, 0, stream, A, fc);
        else hipLaunchKernelGGL((v6_kernel<false, false, false>), grid, block, 0, stream, A, fc);
    }
    return hipGetLastError();
}
