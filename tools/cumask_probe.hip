// Ground truth for hipExtStreamCreateWithCUMask on MI355X: (1) which XCC / SE / CU the blocks of a masked stream land on,
// for several bit patterns; (2) whether two masked streams with disjoint masks really run side by side — a spin kernel that
// owns a whole CU (160 KiB of LDS), timed alone on each mask and on both at once.
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/bin/cumask_probe && tools/bin/cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(uint32_t* out) {
    extern __shared__ unsigned char lds[];
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
    // stay a little so that blocks spread over every enabled CU
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000ull) {}
    if (lds[threadIdx.x] == 123 && out[0] == 0xdeadbeefu) out[1] = 1;
}

__global__ __launch_bounds__(256) void spin_kernel(unsigned long long ticks, uint32_t* sink) {
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) {}
    if (lds[threadIdx.x] == 123 && sink[0] == 0xdeadbeefu) sink[1] = 1;
}

static hipStream_t masked(const std::vector<int>& bits) {
    uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b : bits) words[b / 32] |= 1u << (b % 32);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, 8, words) != hipSuccess) return nullptr;
    return s;
}

int main() {
    uint32_t* d = nullptr;
    CK(hipMalloc(&d, 2 * 4096 * sizeof(uint32_t)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(where_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    struct Pat { const char* name; std::vector<int> bits; };
    std::vector<Pat> pats;
    { std::vector<int> v; for (int i = 0; i < 256; ++i) v.push_back(i); pats.push_back({"all 256 bits", v}); }
    { std::vector<int> v; for (int i = 0; i < 128; ++i) v.push_back(i); pats.push_back({"bits 0..127", v}); }
    { std::vector<int> v; for (int i = 128; i < 256; ++i) v.push_back(i); pats.push_back({"bits 128..255", v}); }
    { std::vector<int> v; for (int i = 0; i < 256; ++i) if ((i / 8) % 2 == 0) v.push_back(i); pats.push_back({"(i/8) even", v}); }
    { std::vector<int> v; for (int i = 0; i < 256; ++i) if ((i / 8) % 2 == 1) v.push_back(i); pats.push_back({"(i/8) odd", v}); }
    { std::vector<int> v; for (int i = 0; i < 32; ++i) v.push_back(i); pats.push_back({"bits 0..31", v}); }
    { std::vector<int> v; for (int i = 0; i < 256; i += 8) v.push_back(i); pats.push_back({"bits 0,8,16,..", v}); }
    { std::vector<int> v; for (int i = 0; i < 256; ++i) if (i % 2 == 0) v.push_back(i); pats.push_back({"even bits", v}); }
    std::vector<uint32_t> h(2 * 4096);
    for (auto& p : pats) {
        hipStream_t s = masked(p.bits);
        if (!s) { printf("%-16s: stream creation failed\n", p.name); continue; }
        CK(hipMemsetAsync(d, 0xff, 2 * 4096 * sizeof(uint32_t), s));
        hipLaunchKernelGGL(where_kernel, dim3(2048), dim3(64), 160 * 1024, s, d);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, 2 * 2048 * sizeof(uint32_t), hipMemcpyDeviceToHost));
        int per_xcc[8] = {0};
        std::set<uint32_t> cus;
        for (int b = 0; b < 2048; ++b) {
            const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
            per_xcc[xcc & 7]++;
            cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
        }
        printf("%-16s: %3zu bits -> %3zu distinct (xcc,se,sh,cu); blocks per XCC:", p.name, p.bits.size(), cus.size());
        for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    // concurrency: 256 CU-filling blocks of ~100 us each, on mask A alone, mask B alone, both at once
    auto time_pair = [&](const std::vector<int>& a, const std::vector<int>& b, const char* name) -> int {
        hipStream_t sa = masked(a), sb = masked(b);
        if (!sa || !sb) { printf("%s: stream creation failed\n", name); return 0; }
        const unsigned long long ticks = 200000ull;   // s_memtime runs at ~1.5-2 GHz here: ~100 us
        auto run = [&](bool ua, bool ub) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 5; ++r) {
                if (ua) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 160 * 1024, sa, ticks, d);
                if (ub) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 160 * 1024, sb, ticks, d);
            }
            CK(hipDeviceSynchronize());
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("  %s: %s%s 5 launches of 256 blocks x 100 us each: %.0f us\n", name, ua ? "A " : "", ub ? "B " : "", us);
            return 0;
        };
        run(true, false); run(true, false); run(false, true); run(true, true); run(true, true);
        CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
        return 0;
    };
    time_pair(pats[1].bits, pats[2].bits, "0..127 | 128..255");
    time_pair(pats[3].bits, pats[4].bits, "(i/8) even | odd  ");
    time_pair(pats[0].bits, pats[0].bits, "all | all         ");
    return 0;
}
