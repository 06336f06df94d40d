// Shared host/device helpers for libwise_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include <mutex>

#include "../../include/wise_hip.h"

namespace wise {

void set_error(const char* fmt, ...);

#define WISE_CHECK_ARG(cond, ...)          \
    do {                                   \
        if (!(cond)) {                     \
            ::wise::set_error(__VA_ARGS__); \
            return WISE_E_INVALID;         \
        }                                  \
    } while (0)

// kernel launches report through hipGetLastError (no sync: graph-capturable)
#define WISE_LAUNCH_CHECK(name)                                                     \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            ::wise::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Function attributes (and __device__ tables) belong to a DEVICE, not to the process: `once(f)` runs f the first time the
// calling thread's current device comes by, so a process that drives several GPUs sets every one of them up.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};   // bit d: device d is set up
    std::mutex mu;
    template <typename F>
    void operator()(F&& f) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> lock(mu);
        if (done.load(std::memory_order_relaxed) & bit) return;
        f();
        done.fetch_or(bit, std::memory_order_release);
    }
};
// Raise a kernel's dynamic-LDS limit on the current device.  A failure is recorded (wise_last_error); the launch behind
// it then fails with its own error, which WISE_LAUNCH_CHECK reports.
static inline void raise_lds_limit(const void* kern, int bytes) {
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) set_error("hipFuncSetAttribute(max dynamic LDS = %d): %s", bytes, hipGetErrorString(e));
}

// Optional per-kernel HIP-event brackets (wise_prof_begin/_end in the C ABI): class 0 = bf16 GEMM
// (work = flop), class 1 = IP scan (work = algorithmic bytes).  No-ops unless profiling is on.
enum : int { PROF_GEMM = 0, PROF_SCAN = 1, PROF_CLASSES = 2 };
int prof_open(int cls, double work, hipStream_t st);   // returns slot or -1
void prof_close(int slot, hipStream_t st);
struct ProfScope {
    int slot; hipStream_t st;
    ProfScope(int cls, double work, hipStream_t s) : slot(prof_open(cls, work, s)), st(s) {}
    ~ProfScope() { if (slot >= 0) prof_close(slot, st); }
};

// ---- bf16 bit helpers (device) -------------------------------------------------------------
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even, NaN kept a NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
// two values in ONE v_cvt_pk_bf16_f32 (the scalar form above costs a conversion per value plus a shift and an or per pair)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(unsigned, v);
}

using short8 = __attribute__((ext_vector_type(8))) short;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// wave64 all-reduce helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace wise
