#include "common.h"

namespace fpcc {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace fpcc

extern "C" const char *fpcc_last_error(void) { return fpcc::g_err; }

extern "C" int fpcc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Diagnostic: the shader clock as a kernel sees it.  One wave spins for `spin_us` microseconds of the constant 100 MHz counter
// (s_memrealtime) and reports how many shader-clock cycles (s_memtime) went by: out[0] = cycles, out[1] = 100 MHz ticks.
// Launched between the launches of a step it shows the clock the power management gives the step at that moment
// (tools/gap_probe.py: why a layer takes longer inside the step than alone).
namespace fpcc { namespace {
__global__ void k_clock_probe(int64_t *out, int spin_ticks) {
    if (threadIdx.x != 0) return;
    const uint64_t t0 = wall_clock64(), c0 = clock64();
    uint64_t t1;
    do { t1 = wall_clock64(); } while (t1 - t0 < (uint64_t)spin_ticks);
    const uint64_t c1 = clock64();
    out[0] = (int64_t)(c1 - c0);
    out[1] = (int64_t)(t1 - t0);
}
} }

extern "C" int fpcc_clock_probe(int64_t *out2, int spin_us, void *stream) {
    if (!out2 || spin_us <= 0 || spin_us > 10000) { fpcc::set_error("clock_probe: bad arguments"); return FPCC_E_ARG; }
    hipLaunchKernelGGL(fpcc::k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, out2, spin_us * 100);
    if (hipGetLastError() != hipSuccess) { fpcc::set_error("clock_probe: launch failed"); return FPCC_E_HIP; }
    return 0;
}
