// Shared helpers for libfpcc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../../include/fpcc_hip.h"

namespace fpcc {

void set_error(const char *fmt, ...);

inline int fail_arg(const char *what) {
    set_error("invalid argument: %s", what);
    return FPCC_E_ARG;
}

inline int check_hip(hipError_t e, const char *where) {
    if (e == hipSuccess) return FPCC_OK;
    set_error("%s: %s", where, hipGetErrorString(e));
    return FPCC_E_HIP;
}

#define FPCC_HIP(expr)                                          \
    do {                                                        \
        int rc__ = ::fpcc::check_hip((expr), #expr);            \
        if (rc__ != FPCC_OK) return rc__;                       \
    } while (0)

#define FPCC_LAUNCHED(name) FPCC_HIP((hipGetLastError()))

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;   // gfx950 wavefront

inline unsigned blocks_for(int64_t n, int threads) { return static_cast<unsigned>((n + threads - 1) / threads); }

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// fpcc_conv_i8_also with one more switch (int_ops.hip): ws_zeroed = the offset-split accumulator in `ws` is already clear
int conv_i8_run(const int8_t *a, int c_in, int lda, const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os, int nbr_bias,
                const int8_t *w, int ldw, const int32_t *zp_comp, const int32_t *bias, const int32_t *slope, const uint32_t *requant_mul,
                const int64_t *zero_point, int shift, int out_bits, void *out, int ldo, int out_pad, int c_out, int64_t n_out,
                const int32_t *row_order, const int32_t *residual, int ld_res, const int32_t *slope2, const fpcc_requant8 *also, int n_also,
                void *ws, int64_t ws_bytes, bool ws_zeroed, void *stream);

// fpcc_time_next_launch: the calling thread's next convolution-family entry point records ev0 on its stream before its first launch
// and ev1 after its last one, inside the one C call (see include/fpcc_hip.h)
struct PendingEvents { void *ev0 = nullptr, *ev1 = nullptr; };
PendingEvents &pending_events();
struct LaunchBracket {
    hipStream_t s;
    void *ev1;
    explicit LaunchBracket(void *stream) : s(as_stream(stream)), ev1(nullptr) {
        PendingEvents &p = pending_events();
        if (p.ev0) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(p.ev0), s);
        ev1 = p.ev1;
        p.ev0 = p.ev1 = nullptr;
    }
    ~LaunchBracket() {
        if (ev1) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(ev1), s);
    }
    LaunchBracket(const LaunchBracket &) = delete;
    LaunchBracket &operator=(const LaunchBracket &) = delete;
};

}  // namespace fpcc
