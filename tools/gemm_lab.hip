// Stand-alone A/B harness for the GEMM kernels (no torch): gemm_w4.h against the product library's kernels on the ViT
// shapes, one process, interleaved rounds, random data, every result checked against a plain fp32 GPU reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 <build.py's flags> tools/gemm_lab.hip -o tools/bin/gemm_lab -ldl
//   tools/bin/gemm_lab [shapes: vit|sq|l14|all] [rounds]
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define W4_STAMPS 1
#include "../wise_amd/csrc/gemm_w4.h"
#include "gemm_w4q_lab.h"

namespace wise { void set_error(const char*, ...) {} }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using wise::bf16_t;

__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    // sum of two uniforms: roughly bell-shaped in [-1, 1), full sign and mantissa variety
    const float u = ((x & 0xffff) + (x >> 16)) * (1.f / 65536.f) - 1.f;
    p[i] = wise::f32_to_bf16(u * scale);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    p[i] = (((x & 0xffff) + (x >> 16)) * (1.f / 65536.f) - 1.f) * scale;
}
// reference rows [r0, r0 + rows): ref[m][n] = sum_k A[m][k] W[n][k] + bias[n]   (fp32 fma chain)
__global__ void ref_gemm(const bf16_t* A, const bf16_t* W, const float* bias, int N, int K, int r0, float* ref) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = r0 + blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(wise::bf16_to_f32(A[(size_t)m * K + k]), wise::bf16_to_f32(W[(size_t)n * K + k]), s);
    ref[(size_t)blockIdx.y * N + n] = s + bias[n];
}

struct Shape { const char* name; int M, N, K, mode; };

typedef int (*gemm_fn)(const uint16_t*, const uint16_t*, const float*, int, int, int, int, void*, void*);
typedef int (*variant_fn)(int);

static float host_act(float x, int mode) {
    if (mode == 1) return x / (1.f + expf(-1.702f * x));
    if (mode == 2) return 0.5f * x * (1.f + erff(x * 0.70710678f));
    return x;
}

template <int MI, int NJ, int SA, int SW, int OCC = 1>
static void launch_new(int mode, const bf16_t* A, const bf16_t* W, const float* b, int M, int N, int K, void* out, hipStream_t st) {
    using namespace wise;
    switch (mode) {
        case 0: launch_w4<EPI_BF16, MI, NJ, SA, SW, OCC>(A, W, b, M, N, K, out, st); break;
        case 1: launch_w4<EPI_QUICKGELU, MI, NJ, SA, SW, OCC>(A, W, b, M, N, K, out, st); break;
        case 2: launch_w4<EPI_GELU, MI, NJ, SA, SW, OCC>(A, W, b, M, N, K, out, st); break;
        case 3: launch_w4<EPI_RESID, MI, NJ, SA, SW, OCC>(A, W, b, M, N, K, out, st); break;
        case 4: launch_w4<EPI_F32, MI, NJ, SA, SW, OCC>(A, W, b, M, N, K, out, st); break;
    }
}

int main(int argc, char** argv) {
    std::string which = argc > 1 ? argv[1] : "vit";
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    const int NB = argc > 3 ? atoi(argv[3]) : 1;   // buffer sets rotated between launches (working set beyond the 256 MB Infinity Cache)
    std::vector<Shape> shapes;
    if (which == "vit" || which == "all") {
        shapes.push_back({"qkv  12800x2304x768", 12800, 2304, 768, 0});
        shapes.push_back({"out  12800x768x768", 12800, 768, 768, 3});
        shapes.push_back({"fc1  12800x3072x768", 12800, 3072, 768, 1});
        shapes.push_back({"fc2  12800x768x3072", 12800, 768, 3072, 3});
        shapes.push_back({"fc1-noact 12800x3072x768", 12800, 3072, 768, 0});
        shapes.push_back({"fc2-f32store 12800x768x3072", 12800, 768, 3072, 4});
    }
    if (which == "htsat" || which == "all") {   // MS-CLAP HTSAT at 128 clips: stages 2 (C=192), 3 (C=384), 4 (C=768)
        shapes.push_back({"s2-qkv 131072x576x192", 131072, 576, 192, 0});
        shapes.push_back({"s2-proj 131072x192x192", 131072, 192, 192, 3});
        shapes.push_back({"s2-fc1 131072x768x192", 131072, 768, 192, 2});
        shapes.push_back({"s2-fc2 131072x192x768", 131072, 192, 768, 3});
        shapes.push_back({"s3-qkv 32768x1152x384", 32768, 1152, 384, 0});
        shapes.push_back({"s3-proj 32768x384x384", 32768, 384, 384, 3});
        shapes.push_back({"s3-fc1 32768x1536x384", 32768, 1536, 384, 2});
        shapes.push_back({"s3-fc2 32768x384x1536", 32768, 384, 1536, 3});
        shapes.push_back({"s4-qkv 8192x2304x768", 8192, 2304, 768, 0});
        shapes.push_back({"s4-proj 8192x768x768", 8192, 768, 768, 3});
        shapes.push_back({"s4-fc1 8192x3072x768", 8192, 3072, 768, 2});
        shapes.push_back({"s4-fc2 8192x768x3072", 8192, 768, 3072, 3});
    }
    if (which == "patch" || which == "all") shapes.push_back({"patch 12544x768x3072", 12544, 768, 3072, 4});
    if (which == "sq" || which == "all") {
        shapes.push_back({"sq4096", 4096, 4096, 4096, 0});
        shapes.push_back({"sq8192", 8192, 8192, 8192, 0});
    }
    if (which == "l14" || which == "all") {
        shapes.push_back({"L14qkv 65792x3072x1024", 65792, 3072, 1024, 0});
        shapes.push_back({"L14out 65792x1024x1024", 65792, 1024, 1024, 3});
        shapes.push_back({"L14fc1 65792x4096x1024", 65792, 4096, 1024, 1});
        shapes.push_back({"L14fc2 65792x1024x4096", 65792, 1024, 4096, 3});
    }
    if (which == "text" || which == "all") {   // CLIP B/32 text tower, 256 queries x 77 tokens
        shapes.push_back({"Tqkv 19712x1536x512", 19712, 1536, 512, 0});
        shapes.push_back({"Tout 19712x512x512", 19712, 512, 512, 3});
        shapes.push_back({"Tfc1 19712x2048x512", 19712, 2048, 512, 1});
        shapes.push_back({"Tfc2 19712x512x2048", 19712, 512, 2048, 3});
    }
    if (which == "xlmr" || which == "all") {   // XLM-RoBERTa-large text tower, 256 queries x 77 tokens
        shapes.push_back({"Xqkv 19712x3072x1024", 19712, 3072, 1024, 0});
        shapes.push_back({"Xout 19712x1024x1024", 19712, 1024, 1024, 3});
        shapes.push_back({"Xfc1 19712x4096x1024", 19712, 4096, 1024, 2});
        shapes.push_back({"Xfc2 19712x1024x4096", 19712, 1024, 4096, 3});
    }
    if (which == "h14" || which == "all") {
        shapes.push_back({"H14qkv 65792x3840x1280", 65792, 3840, 1280, 0});
        shapes.push_back({"H14out 65792x1280x1280", 65792, 1280, 1280, 3});
        shapes.push_back({"H14fc1 65792x5120x1280", 65792, 5120, 1280, 2});
        shapes.push_back({"H14fc2 65792x1280x5120", 65792, 1280, 5120, 3});
    }
    void* h = dlopen("wise_amd/lib/libwise_hip_debug.so", RTLD_NOW | RTLD_LOCAL);
    gemm_fn old_gemm = h ? (gemm_fn)dlsym(h, "wise_gemm_bf16") : nullptr;
    variant_fn set_variant = h ? (variant_fn)dlsym(h, "wise_debug_set_gemm_variant") : nullptr;
    if (!old_gemm) printf("(libwise_hip_debug.so not loaded: %s) new kernels only\n", dlerror());

    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ITERS = 20;

    for (const Shape& s : shapes) {
        const size_t nA = (size_t)s.M * s.K, nW = (size_t)s.N * s.K, nC = (size_t)s.M * s.N;
        bf16_t *A, *W; float* bias; void* out; float* ref;
        const bool f32o = s.mode == 3 || s.mode == 4;
        bf16_t* Abase; unsigned char* outbase;
        const size_t outB = nC * (f32o ? 4 : 2);
        CK(hipMalloc(&Abase, nA * 2 * NB)); CK(hipMalloc(&W, nW * 2)); CK(hipMalloc(&bias, s.N * 4));
        CK(hipMalloc(&outbase, outB * NB));
        A = Abase; out = outbase;
        int rot = 0;
        const int RR = 512;   // reference rows: 2 x 256 rows spread over the matrix
        CK(hipMalloc(&ref, (size_t)RR * s.N * 4));
        for (int b = 0; b < NB; ++b) fill_bf16<<<(nA + 255) / 256, 256, 0, st>>>(Abase + b * nA, nA, 0x1234u, 1.0f);
        fill_bf16<<<(nW + 255) / 256, 256, 0, st>>>(W, nW, 0x9e37u, 2.0f / sqrtf((float)s.K));
        fill_f32<<<(s.N + 255) / 256, 256, 0, st>>>(bias, s.N, 0x77u, 0.5f);
        CK(hipStreamSynchronize(st));

        struct Var { std::string name; int kind; int arg; };   // kind 0: old library variant arg (-1 = auto); 1: w4 MI=arg
        std::vector<Var> vars;
        if (old_gemm) {
            vars.push_back({"lib auto", 0, -1});
        }
        if (wise::w4_shape_ok(s.M, s.N, s.K, 8)) vars.push_back({"w4 256x256", 1, 8});
        if (wise::w4_shape_ok(s.M, s.N, s.K, 5)) vars.push_back({"w4 160x256", 1, 5});
        if (wise::w4_shape_ok(s.M, s.N, s.K, 4)) vars.push_back({"w4 128x256", 1, 4});
        if (wise::w4_shape_ok(s.M, s.N, s.K, 10)) vars.push_back({"w4 320x256", 1, 10});
        if (wise::w4_shape_ok(s.M, s.N, s.K, 10, 6)) vars.push_back({"w4 320x192", 1, 106});
        if (wise::w4_shape_ok(s.M, s.N, s.K, 7, 6)) vars.push_back({"w4 224x192", 1, 76});
        if (wise::w4_shape_ok(s.M, s.N, s.K, 8, 6)) vars.push_back({"w4 256x192", 1, 86});
        if (wise::w4_shape_ok(s.M, s.N, s.K, 4, 6)) vars.push_back({"w4 128x192", 1, 46});
        if (wise::w4_shape_ok(s.M, s.N, s.K, 4, 6)) vars.push_back({"w4 128x192 two per CU", 1, 462});
        if (wise::w4_shape_ok(s.M, s.N, s.K, 4, 4)) vars.push_back({"w4 128x128 two per CU", 1, 442});
        if (wise::w4q_shape_ok(s.M, s.N, s.K, s.mode) && !f32o) vars.push_back({"w4q two-set 128x256", 1, 600});
        if (wise::w4p_shape_ok(s.M, s.N, s.K) && !f32o) vars.push_back({"w4p persistent 160x256", 1, 500});

        auto run = [&](const Var& v) {
            rot = (rot + 1) % NB;
            A = Abase + (size_t)rot * nA; out = outbase + (size_t)rot * outB;
            if (v.kind == 0) {
                int rc = old_gemm(A, W, bias, s.M, s.N, s.K, s.mode, out, st);
                if (rc) { printf("old gemm rc %d\n", rc); exit(1); }
            } else if (v.arg == 8) launch_new<8, 8, 3, 2>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 5) launch_new<5, 8, 3, 2>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 4) launch_new<4, 8, 3, 2>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 10) launch_new<10, 8, 2, 2>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 106) launch_new<10, 6, 2, 3>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 76) launch_new<7, 6, 3, 2>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 86) launch_new<8, 6, 3, 2>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 46) launch_new<4, 6, 3, 2>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 462) launch_new<4, 6, 2, 2, 2>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 442) launch_new<4, 4, 3, 2, 2>(s.mode, A, W, bias, s.M, s.N, s.K, out, st);
            else if (v.arg == 600) {
                using namespace wise;
                if (s.mode == 0) launch_w4q<EPI_BF16>(A, W, bias, s.M, s.N, s.K, out, 256, st);
                else if (s.mode == 1) launch_w4q<EPI_QUICKGELU>(A, W, bias, s.M, s.N, s.K, out, 256, st);
                else launch_w4q<EPI_GELU>(A, W, bias, s.M, s.N, s.K, out, 256, st);
            }
            else if (v.arg == 500) {
                using namespace wise;
                if (s.mode == 0) launch_w4p<EPI_BF16>(A, W, bias, s.M, s.N, s.K, out, 256, st);
                else if (s.mode == 1) launch_w4p<EPI_QUICKGELU>(A, W, bias, s.M, s.N, s.K, out, 256, st);
                else launch_w4p<EPI_GELU>(A, W, bias, s.M, s.N, s.K, out, 256, st);
            }
        };

        // correctness: rows [0,256) and the last 256 rows against the reference
        std::vector<float> href((size_t)RR * s.N), hout_f;
        std::vector<uint16_t> hout_b;
        for (const Var& v : vars) {
            CK(hipMemsetAsync(outbase, 0, outB * NB, st));
            run(v);
            CK(hipStreamSynchronize(st));
            CK(hipGetLastError());
            double maxerr = 0, maxref = 0;
            for (int part = 0; part < 2; ++part) {
                const int r0 = part == 0 ? 0 : s.M - 256;
                ref_gemm<<<dim3((s.N + 255) / 256, 256), 256, 0, st>>>(A, W, bias, s.N, s.K, r0, ref);
                CK(hipMemcpyAsync(href.data(), ref, (size_t)256 * s.N * 4, hipMemcpyDeviceToHost, st));
                if (f32o) {
                    hout_f.resize((size_t)256 * s.N);
                    CK(hipMemcpyAsync(hout_f.data(), (float*)out + (size_t)r0 * s.N, (size_t)256 * s.N * 4, hipMemcpyDeviceToHost, st));
                } else {
                    hout_b.resize((size_t)256 * s.N);
                    CK(hipMemcpyAsync(hout_b.data(), (uint16_t*)out + (size_t)r0 * s.N, (size_t)256 * s.N * 2, hipMemcpyDeviceToHost, st));
                }
                CK(hipStreamSynchronize(st));
                for (size_t i = 0; i < (size_t)256 * s.N; ++i) {
                    const float r = host_act(href[i], s.mode);
                    float o;
                    if (f32o) o = hout_f[i];
                    else { unsigned u = (unsigned)hout_b[i] << 16; memcpy(&o, &u, 4); }
                    const double tol_scale = f32o ? 1.0 : (1.0 + fabs(r)) ;
                    maxerr = std::max(maxerr, fabs((double)o - r) / tol_scale);
                    maxref = std::max(maxref, (double)fabs(r));
                }
            }
            printf("  check %-22s %-20s max err %.3e (max |ref| %.2f) %s\n", s.name, v.name.c_str(), maxerr, maxref,
                   maxerr < (f32o ? 2e-3 : 1.2e-2) ? "ok" : "MISMATCH");
        }
        // timing: interleaved rounds
        std::vector<std::vector<float>> us(vars.size());
        for (int r = 0; r < rounds; ++r)
            for (size_t vi = 0; vi < vars.size(); ++vi) {
                for (int w = 0; w < 3; ++w) run(vars[vi]);
                CK(hipEventRecord(e0, st));
                for (int it = 0; it < ITERS; ++it) run(vars[vi]);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                us[vi].push_back(ms * 1000.f / ITERS);
            }
        const double flop = 2.0 * s.M * s.N * s.K;
        for (size_t vi = 0; vi < vars.size(); ++vi) {
            std::sort(us[vi].begin(), us[vi].end());
            const float med = us[vi][us[vi].size() / 2], mn = us[vi][0];
            printf("%-24s %-20s median %8.2f us  %7.1f TFLOP/s   (min %8.2f us %7.1f)\n", s.name, vars[vi].name.c_str(), med,
                   flop / med * 1e-6, mn, flop / mn * 1e-6);
        }
        for (size_t vi = 0; vi < vars.size(); ++vi) {
            if (vars[vi].kind == 1 && vars[vi].arg == 600) {
                run(vars[vi]); run(vars[vi]);
                CK(hipStreamSynchronize(st));
                unsigned long long hp[4][4];
                CK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(wise::w4::g_w4p_stamps), sizeof(hp)));
                for (int t = 0; t < 3; ++t)
                    printf("  stamps w4q tile %d: drain steps %6llu  rest of loop %6llu  tile end %6llu\n", t,
                           hp[t][1] - hp[t][0], hp[t][2] - hp[t][1], hp[t][3] - hp[t][2]);
                continue;
            }
            if (vars[vi].kind == 1 && vars[vi].arg == 500) {
                run(vars[vi]); run(vars[vi]);
                CK(hipStreamSynchronize(st));
                unsigned long long hp[4][4];
                CK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(wise::w4::g_w4p_stamps), sizeof(hp)));
                for (int t = 0; t < 3; ++t)
                    printf("  stamps w4p tile %d: steps 0-4 %6llu  rest of loop %6llu  pack %6llu  (to next tile start %6llu)\n", t,
                           hp[t][1] - hp[t][0], hp[t][2] - hp[t][1], hp[t][3] - hp[t][2], t < 2 ? hp[t + 1][0] - hp[t][3] : 0ull);
                continue;
            }
            if (vars[vi].kind != 1) continue;
            run(vars[vi]); run(vars[vi]);
            CK(hipStreamSynchronize(st));
            unsigned long long hs[2][4];
            CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(wise::w4::g_w4_stamps), sizeof(hs)));
            for (int b = 0; b < 2; ++b)
                printf("  stamps %-20s %s block: prologue %6llu  loop %7llu  epilogue %6llu cycles (s_memtime, 100 MHz ticks x?)\n",
                       vars[vi].name.c_str(), b ? "last " : "first", hs[b][1] - hs[b][0], hs[b][2] - hs[b][1], hs[b][3] - hs[b][2]);
        }
        fflush(stdout);
        CK(hipFree(Abase)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(outbase)); CK(hipFree(ref));
    }
    return 0;
}
