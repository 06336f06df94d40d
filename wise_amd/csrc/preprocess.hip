// the image transform in front of the tower, on the GPU.
//
// Replaces, for uint8 frames [n,3,H,W] as the decoder hands them over (src/dataloader/dataset.py:298), the
// per-frame CPU loop of src/feature/mlfoundation_openclip.py:81-90:
//     F.to_pil_image -> Resize(S, BICUBIC, shorter side) -> CenterCrop(S) -> [ToTensor -> Normalize]
// The bracketed part is already fused into the patch gather of the tower (vit.hip, WISE_VIT_IN_U8); this file
// produces the uint8 [n,3,S,S] crop.  The arithmetic is Pillow's 8-bit resampler (libImaging/Resample.c), which
// is exact integer work and is reproduced bit for bit:
//   * taps per output position from `precompute_coeffs` (double; support = 2 * max(scale, 1): antialiased),
//     rounded to 22-bit fixed point by `normalize_coeffs_8bpc`                         [host, plan_init]
//   * horizontal pass over the input rows, each result rounded to uint8 (`clip8`), then the vertical pass over
//     that uint8 image, rounded again                                                   [device, fused]
// Only the S x S crop is produced, so only the columns/rows it depends on are ever read.
//
// Kernel: one workgroup per (plane, TS x TS output tile).  It stages the input rectangle the tile depends on in
// LDS (dword loads when W % 4 == 0), runs the horizontal pass LDS -> LDS and the vertical pass LDS -> HBM.
// Bound: HBM (input bytes read once + halo, S*S bytes written per plane); VALU work is ~2 * taps MACs / output.
//   * every LDS data access is a dword holding 4 taps' bytes: the tap rows are stored on the host already
//     shifted by (first input index & 3) and zero-padded to whole dwords, so the kernel has no alignment
//     cases — a zero coefficient annihilates whatever byte sits under it;
//   * a lane owns 4 rows (horizontal) / 4 columns (vertical) and shares one 16-byte coefficient read between
//     them; rows/columns are stored 4-way interleaved with an odd dword stride so lanes hit distinct banks;
//   * tiles of one plane are consecutive on one XCD (block id -> XCD is round-robin), so the halo a tile
//     shares with its neighbours is served by that XCD's L2.
#include <math.h>

#include <algorithm>
#include <vector>

#include "common.h"

// host: Pillow's tap tables.  Floating point must not be contracted into FMAs: the products and sums below
// are the ones Resample.c performs, in its order.
#pragma clang fp contract(off)

namespace wise {

constexpr int PREC = 32 - 8 - 2;  // Resample.c PRECISION_BITS for 8-bit channels

static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

struct Taps {
    int ksize = 0;
    std::vector<int> first, count, coef;  // coef[out][ksize]
};

static void pillow_taps(int in_size, int out_size, Taps& t) {
    t.first.assign(out_size, 0);
    t.count.assign(out_size, 0);
    if (in_size == out_size) {  // Pillow skips a pass whose size does not change: identity taps
        t.ksize = 1;
        t.coef.assign(out_size, 1 << PREC);
        for (int i = 0; i < out_size; ++i) { t.first[i] = i; t.count[i] = 1; }
        return;
    }
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    t.ksize = (int)ceil(support) * 2 + 1;
    t.coef.assign((size_t)out_size * t.ksize, 0);
    std::vector<double> k(t.ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            t.coef[(size_t)xx * t.ksize + x] =
                k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PREC)) : (int)(0.5 + k[x] * (1 << PREC));
        }
        t.first[xx] = xmin;
        t.count[xx] = xmax;
    }
}

// torchvision CenterCrop: int(round(v / 2.0)) with Python's round-half-to-even
static int half_round_even(int v) {
    const int f = v / 2;
    if ((v & 1) == 0) return f;
    return (f & 1) ? f + 1 : f;
}

struct AxisTables {
    int nd = 0;                      // dwords of padded taps per output
    std::vector<int> start;          // [S] first input dword
    std::vector<int> coef;           // [S][nd*4] taps shifted by (first & 3), zero padded
    std::vector<int> packed;         // [S][nd][4] the same taps as byte planes (device form)
    std::vector<int> t0, tn;         // per tile: first staged dword, number of staged dwords
    int max_n = 0;
};

static void axis_tables(const Taps& t, int off, int S, int tile, int in_size, AxisTables& a) {
    a.nd = 1;
    for (int i = 0; i < S; ++i) a.nd = std::max(a.nd, ((t.first[i + off] & 3) + t.count[i + off] + 3) / 4);
    a.start.resize(S);
    a.coef.assign((size_t)S * a.nd * 4, 0);
    for (int i = 0; i < S; ++i) {
        const int f = t.first[i + off];
        a.start[i] = f >> 2;
        for (int k = 0; k < t.count[i + off]; ++k)
            a.coef[(size_t)i * a.nd * 4 + (f & 3) + k] = t.coef[(size_t)(i + off) * t.ksize + k];
    }
    // device form: tap c is kept as c + 2^22 (>= 0, < 2^24) split into three byte planes, so that four taps are
    // one v_dot4_u32_u8 per plane; per output and dword j: {plane2, plane1, plane0, 0}
    a.packed.assign((size_t)S * a.nd * 4, 0);
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < a.nd; ++j) {
            unsigned pl[3] = {0, 0, 0};
            for (int b = 0; b < 4; ++b) {
                const unsigned c = (unsigned)(a.coef[((size_t)i * a.nd + j) * 4 + b] + (1 << PREC));
                pl[0] |= ((c >> 16) & 255u) << (8 * b);
                pl[1] |= ((c >> 8) & 255u) << (8 * b);
                pl[2] |= (c & 255u) << (8 * b);
            }
            for (int q = 0; q < 3; ++q) a.packed[((size_t)i * a.nd + j) * 4 + q] = (int)pl[q];
        }
    const int nt = (S + tile - 1) / tile;
    a.t0.resize(nt);
    a.tn.resize(nt);
    a.max_n = 0;
    (void)in_size;
    for (int ti = 0; ti < nt; ++ti) {
        const int lo = ti * tile, hi = std::min(S, lo + tile);
        int d0 = a.start[lo], d1 = 0;
        for (int i = lo; i < hi; ++i) { d0 = std::min(d0, a.start[i]); d1 = std::max(d1, a.start[i] + a.nd); }
        a.t0[ti] = d0;
        a.tn[ti] = d1 - d0;
        a.max_n = std::max(a.max_n, d1 - d0);
    }
}

// ---- the matrix-core form (round 4) -------------------------------------------------------------------------------
// A separable resampling pass is a banded matrix product: T[r, x] = sum_c I[r, c] * tap[x, c].  Sixteen outputs depend on a
// window of (16 * scale + 2 * support) consecutive inputs, which v_mfma_i32_16x16x64_i8 covers in one or a few 64-deep
// steps: the data bytes are one operand exactly as they lie in LDS (as d - 128: signed), the taps the other, as three
// SIGNED byte planes tap = c2 * 65536 + c1 * 256 + c0 (balanced digits), so a pass is 3 * ceil(window / 64) MFMAs per
// 16 x 16 outputs instead of ~1400 vector instructions per 32 x 32 tile; sum d * tap = sum (d - 128) * tap + 128 * sum tap,
// the last term a per-output constant folded into the rounding bias.  Integer arithmetic throughout: bit-identical to the
// dot-product kernel and to Pillow.  Per axis: windows start on multiples of 16 (aligned 16-byte operand reads from LDS).
struct MfmaAxis {
    int ng = 0, nkb = 0, max_tn = 0;
    std::vector<int> ws;            // [ng] first input index of a group's window (multiple of 16)
    std::vector<int> bias;          // [ng*16] 128 * sum of taps + 2^21 (zero past S)
    std::vector<int> t0, tn;        // per 32-output tile: first staged input index, staged count (multiples of 16)
    std::vector<unsigned char> bands;   // [ng][nkb][3][64 lanes][16 bytes] in operand order (filled on request)
};

static void mfma_axis(const Taps& t, int off, int S, MfmaAxis& a, bool fill) {
    a.ng = (S + 15) / 16;
    a.ws.assign(a.ng, 0);
    std::vector<int> end(a.ng, 0);
    a.nkb = 1;
    for (int g = 0; g < a.ng; ++g) {
        int lo = 1 << 30, hi = 0;
        for (int o = 16 * g; o < std::min(S, 16 * g + 16); ++o) {
            lo = std::min(lo, t.first[o + off]);
            hi = std::max(hi, t.first[o + off] + t.count[o + off]);
        }
        a.ws[g] = lo & ~15;
        end[g] = hi;
        a.nkb = std::max(a.nkb, (hi - a.ws[g] + 63) / 64);
    }
    a.bias.assign((size_t)a.ng * 16, 0);
    for (int o = 0; o < S; ++o) {
        long long sum = 0;
        for (int k = 0; k < t.count[o + off]; ++k) sum += t.coef[(size_t)(o + off) * t.ksize + k];
        a.bias[o] = (int)(128 * sum + (1 << (PREC - 1)));
    }
    const int nt = (S + 31) / 32;
    a.t0.assign(nt, 0);
    a.tn.assign(nt, 0);
    a.max_tn = 0;
    for (int ti = 0; ti < nt; ++ti) {
        const int g0 = 2 * ti, g1 = std::min(a.ng - 1, 2 * ti + 1);
        const int lo = std::min(a.ws[g0], a.ws[g1]);
        const int hi = std::max(a.ws[g0], a.ws[g1]) + 64 * a.nkb;
        a.t0[ti] = lo;
        a.tn[ti] = hi - lo;
        a.max_tn = std::max(a.max_tn, hi - lo);
    }
    if (!fill) return;
    a.bands.assign((size_t)a.ng * a.nkb * 3 * 1024, 0);
    for (int g = 0; g < a.ng; ++g)
        for (int lane = 0; lane < 64; ++lane) {
            const int o = 16 * g + (lane & 15);
            if (o >= S) continue;
            for (int kb = 0; kb < a.nkb; ++kb)
                for (int b = 0; b < 16; ++b) {
                    const int in = a.ws[g] + kb * 64 + (lane >> 4) * 16 + b;
                    const int k = in - t.first[o + off];
                    if (k < 0 || k >= t.count[o + off]) continue;
                    const int tap = t.coef[(size_t)(o + off) * t.ksize + k];
                    const int c0 = ((tap + 128) & 255) - 128;
                    const int t1 = (tap - c0) >> 8;
                    const int c1 = ((t1 + 128) & 255) - 128;
                    const int c2 = (t1 - c1) >> 8;
                    const int d[3] = {c2, c1, c0};
                    for (int pl = 0; pl < 3; ++pl)
                        a.bands[((((size_t)g * a.nkb + kb) * 3 + pl) * 64 + lane) * 16 + b] = (unsigned char)(signed char)d[pl];
                }
        }
}

struct HostPlan {
    wise_preproc_plan p;
    AxisTables h, v;
    // matrix-core form: available when mfma_ok (moderate scales: up to four 64-deep steps per 16 outputs, 64 KiB of LDS per tile)
    bool mfma_ok = false;
    MfmaAxis mh, mv;
    int m_rstr = 0, m_tstr = 0, m_lds = 0;
    size_t m_ints = 0;      // int32 part of the appended tables; the band bytes follow
};
constexpr int PREPROC_MFMA_FLAG = 2;   // wise_preproc_plan.reserved bit 1: the plan's table blob carries the matrix-core tables
constexpr int PREPROC_MFMA4_FLAG = 4;  // bit 2 (set by the caller): the matrix-core kernel with four waves per tile instead of one

static int build_plan(int H, int W, int S, int squash, HostPlan& hp, bool fill_bands = false) {
    WISE_CHECK_ARG(H >= 1 && W >= 1 && H <= 16384 && W <= 16384, "preproc: frame %dx%d out of range", H, W);
    WISE_CHECK_ARG(S >= 4 && S <= 1024 && S % 4 == 0, "preproc: output edge %d must be a multiple of 4 in [4,1024]", S);
    wise_preproc_plan& p = hp.p;
    p = wise_preproc_plan{};
    p.H = H; p.W = W; p.S = S;
    p.reserved = squash ? 1 : 0;
    if (squash) {
        // open_clip resize_mode 'squash' (the SigLIP models): Resize((S, S)) without regard to aspect, no crop
        p.new_w = S; p.new_h = S; p.left = 0; p.top = 0;
    } else {
        // torchvision Resize(S): shorter side -> S, longer side -> int(S * long / short)
        if (W <= H) { p.new_w = S; p.new_h = (int)((double)((long long)S * H) / (double)W); }
        else        { p.new_w = (int)((double)((long long)S * W) / (double)H); p.new_h = S; }
        WISE_CHECK_ARG(p.new_w <= 65536 && p.new_h <= 65536, "preproc: aspect ratio of %dx%d too extreme", H, W);
        p.left = half_round_even(p.new_w - S);
        p.top = half_round_even(p.new_h - S);
    }
    Taps th, tv;
    pillow_taps(W, p.new_w, th);
    pillow_taps(H, p.new_h, tv);
    for (int tile : {32, 16, 8}) {
        if (tile > S && tile != 8) continue;
        axis_tables(th, p.left, S, tile, W, hp.h);
        axis_tables(tv, p.top, S, tile, H, hp.v);
        p.tile = tile;
        p.ndh = hp.h.nd; p.ndv = hp.v.nd;
        p.max_cols4 = hp.h.max_n; p.max_rows4 = hp.v.max_n;
        const long long cs = p.max_cols4 | 1, rs = p.max_rows4 | 1;
        const long long words = (long long)p.max_rows4 * 4 * cs + (long long)tile * rs +
                                (long long)tile * (p.ndh + p.ndv) * 4 + 2 * tile;
        p.lds_bytes = (int)std::min<long long>(words * 4, 1 << 30);
        if (words * 4 <= 64 * 1024) break;
    }
    WISE_CHECK_ARG(p.lds_bytes <= 64 * 1024, "preproc: %dx%d -> %d needs %d B of LDS per tile (downscale too large)", H,
                   W, S, p.lds_bytes);
    const int nt = (S + p.tile - 1) / p.tile;
    p.table_bytes = (uint64_t)4 * ((size_t)2 * S + (size_t)S * (p.ndh + p.ndv) * 4 + (size_t)4 * nt);
    // the matrix-core form, where it fits
    mfma_axis(th, p.left, S, hp.mh, fill_bands);
    mfma_axis(tv, p.top, S, hp.mv, fill_bands);
    hp.m_rstr = hp.mh.max_tn + 16;
    hp.m_tstr = hp.mv.max_tn + 16;
    hp.m_lds = hp.mv.max_tn * hp.m_rstr + 32 * hp.m_tstr;
    hp.mfma_ok = hp.mh.nkb <= 4 && hp.mv.nkb <= 4 && hp.m_lds <= 64 * 1024;
    if (hp.mfma_ok) {
        const int nt32 = (S + 31) / 32;
        hp.m_ints = (size_t)hp.mh.ng + hp.mv.ng + (size_t)16 * (hp.mh.ng + hp.mv.ng) + (size_t)4 * nt32;
        hp.m_ints = (hp.m_ints + 3) & ~(size_t)3;     // band bytes start 16-byte aligned
        p.table_bytes += 4 * hp.m_ints + (size_t)(hp.mh.ng * hp.mh.nkb + hp.mv.ng * hp.mv.nkb) * 3 * 1024;
        p.reserved |= PREPROC_MFMA_FLAG;
    }
    return WISE_OK;
}

}  // namespace wise

using namespace wise;

extern "C" int wise_preproc_plan_init(int H, int W, int S, wise_preproc_plan* plan) {
    WISE_CHECK_ARG(plan, "preproc: null plan");
    HostPlan hp;
    const int rc = build_plan(H, W, S, 0, hp);
    if (rc) return rc;
    *plan = hp.p;
    return WISE_OK;
}

extern "C" int wise_preproc_plan_init_squash(int H, int W, int S, wise_preproc_plan* plan) {
    WISE_CHECK_ARG(plan, "preproc: null plan");
    HostPlan hp;
    const int rc = build_plan(H, W, S, 1, hp);
    if (rc) return rc;
    *plan = hp.p;
    return WISE_OK;
}

// blob (int32): hstart[S] | vstart[S] | hplanes[S][ndh][4] | vplanes[S][ndv][4] | hc0[nt] | hcn[nt] | vr0[nt] | vrn[nt]
extern "C" int wise_preproc_tables(const wise_preproc_plan* plan, void* host_tables) {
    WISE_CHECK_ARG(plan && host_tables, "preproc: null argument");
    HostPlan hp;
    const int rc = build_plan(plan->H, plan->W, plan->S, plan->reserved & 1, hp, true);
    if (rc) return rc;
    WISE_CHECK_ARG(hp.p.table_bytes == plan->table_bytes && hp.p.tile == plan->tile, "preproc: plan does not match");
    int* o = static_cast<int*>(host_tables);
    auto put = [&](const std::vector<int>& v) { std::copy(v.begin(), v.end(), o); o += v.size(); };
    put(hp.h.start); put(hp.v.start); put(hp.h.packed); put(hp.v.packed);
    put(hp.h.t0); put(hp.h.tn); put(hp.v.t0); put(hp.v.tn);
    if (hp.mfma_ok) {
        // appended: hws[ngh] | vws[ngv] | hbias[16 ngh] | vbias[16 ngv] | ht0 | htn | vt0 | vtn [nt32 each] | pad to 4 ints |
        //           h bands [ngh][nkbh][3][64][16 bytes] | v bands
        int* base = o;
        put(hp.mh.ws); put(hp.mv.ws); put(hp.mh.bias); put(hp.mv.bias);
        put(hp.mh.t0); put(hp.mh.tn); put(hp.mv.t0); put(hp.mv.tn);
        while ((size_t)(o - base) < hp.m_ints) *o++ = 0;
        unsigned char* ob = reinterpret_cast<unsigned char*>(o);
        std::copy(hp.mh.bands.begin(), hp.mh.bands.end(), ob);
        std::copy(hp.mv.bands.begin(), hp.mv.bands.end(), ob + hp.mh.bands.size());
    }
    return WISE_OK;
}

// Pillow's raw tap table for one axis (tests pin it against the oracle without a GPU).
extern "C" int wise_preproc_taps(int in_size, int out_size, int* ksize, int* first, int* count, int* coef,
                                 int coef_capacity) {
    WISE_CHECK_ARG(in_size >= 1 && out_size >= 1 && ksize && first && count && coef, "preproc_taps: bad argument");
    Taps t;
    pillow_taps(in_size, out_size, t);
    WISE_CHECK_ARG((long long)out_size * t.ksize <= coef_capacity, "preproc_taps: need %lld coefficients",
                   (long long)out_size * t.ksize);
    *ksize = t.ksize;
    std::copy(t.first.begin(), t.first.end(), first);
    std::copy(t.count.begin(), t.count.end(), count);
    std::copy(t.coef.begin(), t.coef.end(), coef);
    return WISE_OK;
}

namespace wise {

// Resample.c clip8: arithmetic shift by PRECISION_BITS, clamp to 0..255.  The shift is kept opaque: left to
// itself the compiler fuses "clamp two shifted ints and pack them" into v_ashr_pk_u8_i32, which on gfx950
// writes only the low half of its destination while the code that follows ORs the whole register (measured:
// bytes 2 and 3 of every packed dword came out OR-ed with stale accumulator bits).
__device__ __forceinline__ int clip8(int acc) {
    int v;
    asm("v_ashrrev_i32 %0, 22, %1" : "=v"(v) : "v"(acc));
    static_assert(PREC == 22, "shift literal above");
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Four taps at once: the data dword's bytes against the three byte planes of (tap + 2^22), plus the byte sum
// that takes the 2^22 offset out again.  All arithmetic is mod 2^32; the true sum fits an int (Resample.c
// accumulates in one), so the wrapped result is the true result.
struct TapAcc {
    unsigned p2 = 0, p1 = 0, p0 = 0, ps = 0;
    __device__ __forceinline__ void add(unsigned w, const int4& c) {
        p2 = __builtin_amdgcn_udot4(w, (unsigned)c.x, p2, false);
        p1 = __builtin_amdgcn_udot4(w, (unsigned)c.y, p1, false);
        p0 = __builtin_amdgcn_udot4(w, (unsigned)c.z, p0, false);
        ps = __builtin_amdgcn_udot4(w, 0x01010101u, ps, false);
    }
    __device__ __forceinline__ unsigned byte() const {
        const int acc = (int)((p2 << 16) + (p1 << 8) + p0 - (ps << PREC) + (1u << (PREC - 1)));
        return (unsigned)clip8(acc);
    }
};

// idx / d for idx < 2^16, d < 2^12 (m = floor((2^32-1)/d) + 1)
__device__ __forceinline__ int fast_div(int idx, int d, unsigned m) { return d == 1 ? idx : (int)__umulhi((unsigned)idx, m); }

template <bool ALIGNED>
__global__ __launch_bounds__(256) void clip_resize_kernel(const unsigned char* __restrict__ frames, int planes, int H,
                                                          int W, int S, int TS, int ndh, int ndv, int CS, int RS,
                                                          int max_rows4, const int* __restrict__ tab,
                                                          unsigned char* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nt = (S + TS - 1) / TS, tiles = nt * nt;
    // blocks b, b+8, b+16 ... share an XCD: give each plane's tiles to one XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int plane = (slot / tiles) * 8 + xcd;
    if (plane >= planes) return;
    const int t = slot % tiles, ty = t / nt, tx = t % nt;

    const int* hstart = tab;
    const int* vstart = tab + S;
    const int* hcoef = tab + 2 * S;
    const int* vcoef = hcoef + (size_t)S * ndh * 4;
    const int* tl = vcoef + (size_t)S * ndv * 4;
    const int c0 = tl[tx], cn = tl[nt + tx], r0 = tl[2 * nt + ty], rn4 = tl[3 * nt + ty];

    unsigned* in32 = reinterpret_cast<unsigned*>(smem);                   // [max_rows4*4][CS], rows 4-way interleaved
    unsigned* tmp32 = in32 + (size_t)max_rows4 * 4 * CS;                  // [TS][RS], columns 4-way interleaved
    int* hco = reinterpret_cast<int*>(tmp32 + (size_t)TS * RS);           // [TS][ndh][4]
    int* vco = hco + TS * ndh * 4;                                        // [TS][ndv][4]
    int* hst = vco + TS * ndv * 4;                                        // [TS]
    int* vst = hst + TS;                                                  // [TS]

    const int tid = threadIdx.x;
    const unsigned char* src = frames + (size_t)plane * H * W;
    const int x0 = tx * TS, y0 = ty * TS;

    // ---- stage the input rectangle (rows r0*4 .. +rn4*4, dwords c0 .. +cn) and the tile's taps
    const int rows = rn4 * 4;
    const int wd = ALIGNED ? (W >> 2) : ((W + 3) >> 2);
    const int cload = min(cn, wd - c0);  // dwords that exist in the frame
    const unsigned m_cload = 0xFFFFFFFFu / (unsigned)cload + 1u;
    for (int idx = tid; idx < rows * cload; idx += 256) {
        const int r = fast_div(idx, cload, m_cload), j = idx - r * cload;
        const int gr = min(r0 * 4 + r, H - 1);  // rows past the frame only ever meet zero taps
        unsigned v;
        if (ALIGNED) {
            v = *reinterpret_cast<const unsigned*>(src + (size_t)gr * W + (size_t)(c0 + j) * 4);
        } else {
            const unsigned char* p = src + (size_t)gr * W;
            const int c = (c0 + j) * 4;
            v = (unsigned)p[c];
            if (c + 1 < W) v |= (unsigned)p[c + 1] << 8;
            if (c + 2 < W) v |= (unsigned)p[c + 2] << 16;
            if (c + 3 < W) v |= (unsigned)p[c + 3] << 24;
        }
        in32[((r & 3) * rn4 + (r >> 2)) * CS + j] = v;
    }
    {
        const int nh = min(TS, S - x0) * ndh * 4, nv = min(TS, S - y0) * ndv * 4;
        for (int idx = tid; idx < TS * ndh * 4; idx += 256) hco[idx] = idx < nh ? hcoef[(size_t)x0 * ndh * 4 + idx] : 0;
        for (int idx = tid; idx < TS * ndv * 4; idx += 256) vco[idx] = idx < nv ? vcoef[(size_t)y0 * ndv * 4 + idx] : 0;
    }
    if (tid < TS) {
        hst[tid] = (x0 + tid < S) ? hstart[x0 + tid] - c0 : 0;
        vst[tid] = (y0 + tid < S) ? vstart[y0 + tid] - r0 : 0;
    }
    __syncthreads();

    // ---- horizontal pass: item (x, rg) = output column x of input rows 4*rg .. 4*rg+3
    const int XG = TS >> 2, xg_shift = 31 - __clz(XG);
    const unsigned m_rn4 = 0xFFFFFFFFu / (unsigned)rn4 + 1u;
    for (int idx = tid; idx < TS * rn4; idx += 256) {
        const int x = fast_div(idx, rn4, m_rn4), rg = idx - x * rn4;
        const int4* co = reinterpret_cast<const int4*>(hco + x * ndh * 4);
        const unsigned* d0 = in32 + rg * CS + hst[x];
        const int rstride = rn4 * CS;
        TapAcc a0, a1, a2, a3;
        for (int j = 0; j < ndh; ++j) {
            const int4 c = co[j];
            a0.add(d0[j], c);
            a1.add(d0[rstride + j], c);
            a2.add(d0[2 * rstride + j], c);
            a3.add(d0[3 * rstride + j], c);
        }
        tmp32[((x & 3) * XG + (x >> 2)) * RS + rg] = a0.byte() | (a1.byte() << 8) | (a2.byte() << 16) | (a3.byte() << 24);
    }
    __syncthreads();

    // ---- vertical pass: item (y, xg) = output row y of columns 4*xg .. 4*xg+3, one dword store
    for (int idx = tid; idx < TS * XG; idx += 256) {
        const int y = idx >> xg_shift, xg = idx & (XG - 1);
        if (y0 + y >= S || x0 + xg * 4 >= S) continue;
        const int4* co = reinterpret_cast<const int4*>(vco + y * ndv * 4);
        const unsigned* d0 = tmp32 + xg * RS + vst[y];
        const int cstride = XG * RS;
        TapAcc a0, a1, a2, a3;
        for (int j = 0; j < ndv; ++j) {
            const int4 c = co[j];
            a0.add(d0[j], c);
            a1.add(d0[cstride + j], c);
            a2.add(d0[2 * cstride + j], c);
            a3.add(d0[3 * cstride + j], c);
        }
        *reinterpret_cast<unsigned*>(out + ((size_t)plane * S + (y0 + y)) * S + x0 + xg * 4) =
            a0.byte() | (a1.byte() << 8) | (a2.byte() << 16) | (a3.byte() << 24);
    }
}


// ------------------------------------------------------------------------------------------------
// The matrix-core form: one workgroup of ONE or FOUR waves per (plane, 32 x 32 output tile) — small frames are served best by
// lone waves (nothing to synchronise, many tiles in flight per CU), large downscales by four waves sharing a tile's staging;
// which of the three kernels (these two, the dot-product one) a geometry takes is measured once per plan by the caller.
//   1. the input rectangle the tile depends on -> LDS as bytes d - 128 (rows: the vertical windows of the tile's two
//      16-row groups; columns: the horizontal windows of its two 16-column groups; 16 lanes cover 64 bytes of a row);
//   2. horizontal pass, per 16 input rows x 16 output columns: D[row][x] = sum_c data[row][c] * tap[x][c] — A = 16 bytes of a
//      row per lane straight from LDS, B = the column group's band planes (operand order, from the plan's tables, held in
//      registers across the row tiles); a lane ends up with 4 consecutive ROWS of one column: rounded, clamped, packed,
//      one ds_write_b32 into the transposed intermediate T^T[x][row] (again as value - 128);
//   3. vertical pass, per 16 output columns x 16 output rows: D[x][y] = sum_r T^T[x][r] * tap[y][r] — a lane ends up with 4
//      consecutive COLUMNS of one output row: one dword store.
// ------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned combine4(const v4i& p2, const v4i& p1, const v4i& p0, int bias, unsigned flip) {
    unsigned w = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int acc = p2[r] * 65536 + p1[r] * 256 + p0[r] + bias;
        w |= (unsigned)clip8(acc) << (8 * r);
    }
    return w ^ flip;
}

template <bool ALIGNED, int NKB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void clip_resize_mfma_kernel(const unsigned char* __restrict__ frames, int planes, int H, int W,
                                                              int S, int nkbh, int nkbv, int RSTR, int TSTR, int tt_off,
                                                              const int* __restrict__ mtab, int ngh, int ngv,
                                                              const unsigned char* __restrict__ hbands,
                                                              const unsigned char* __restrict__ vbands,
                                                              unsigned char* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* rect = smem;
    unsigned char* tt = smem + tt_off;
    const int nt = (S + 31) >> 5, tiles = nt * nt;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int plane = (slot / tiles) * 8 + xcd;
    if (plane >= planes) return;
    const int t = slot % tiles, ty = t / nt, tx = t % nt;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, gq = lane >> 4;
    const int wave = WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int* hws = mtab;
    const int* vws = hws + ngh;
    const int* hbias = vws + ngv;
    const int* vbias = hbias + 16 * ngh;
    const int* tl = vbias + 16 * ngv;
    const int c0 = tl[tx], cn = tl[nt + tx], r0 = tl[2 * nt + ty], rn = tl[3 * nt + ty];
    const unsigned char* src = frames + (size_t)plane * H * W;

    // the band planes of this wave's work, requested first so that they arrive under the staging: horizontal — both column
    // groups (the wave takes row tiles wave, wave + 4, ...); vertical — its one (row group, column group) pair
    v4i BH[2][NKB][3], BV[NKB][3];
    int goff[2], hb[2];
#pragma unroll
    for (int og = 0; og < 2; ++og) {
        const int g = min(2 * tx + og, ngh - 1);
        goff[og] = hws[g] - c0;
        hb[og] = hbias[g * 16 + l15];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                BH[og][kb][pl] = kb < nkbh ? *reinterpret_cast<const v4i*>(hbands + ((((size_t)g * nkbh + kb) * 3 + pl) * 64 + lane) * 16)
                                           : v4i{0, 0, 0, 0};
    }
    int gy = min(2 * ty + (wave >> 1), ngv - 1);
    auto load_bv = [&]() {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                BV[kb][pl] = kb < nkbv ? *reinterpret_cast<const v4i*>(vbands + ((((size_t)gy * nkbv + kb) * 3 + pl) * 64 + lane) * 16)
                                       : v4i{0, 0, 0, 0};
    };
    if (WAVES == 4) load_bv();

    // 1. stage: 16 threads cover 64 bytes of a row, the block 4 * WAVES rows per trip; U loads in flight per thread before
    //    the first goes to LDS (one load per trip leaves every trip waiting out a memory latency on its own)
    constexpr int RPT = 4 * WAVES, U = WAVES == 1 ? 8 : 4;
    const int rq = tid >> 4;
    for (int jc = 0; jc < ((cn + 63) >> 6); ++jc) {
        const int lc = 64 * jc + 4 * l15, gc = c0 + lc;
        for (int rb = 0; rb < rn; rb += RPT * U) {
            unsigned v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int gr = min(r0 + rb + RPT * u + rq, H - 1);         // rows / columns past the frame only ever meet zero taps
                if (ALIGNED) {
                    v[u] = *reinterpret_cast<const unsigned*>(src + (size_t)gr * W + (size_t)min(gc, W - 4));
                } else {
                    const unsigned char* p = src + (size_t)gr * W;
                    v[u] = (unsigned)p[min(gc, W - 1)] | ((unsigned)p[min(gc + 1, W - 1)] << 8) |
                           ((unsigned)p[min(gc + 2, W - 1)] << 16) | ((unsigned)p[min(gc + 3, W - 1)] << 24);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = rb + RPT * u + rq;
                if (r < rn && lc < cn) *reinterpret_cast<unsigned*>(rect + r * RSTR + lc) = v[u] ^ 0x80808080u;
            }
        }
    }
    auto sync = [&]() {
        if (WAVES == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else
            __syncthreads();
    };
    sync();

    // 2. horizontal pass: row tiles wave, wave + WAVES, ...
    for (int mt = wave; mt < (rn >> 4); mt += WAVES) {
#pragma unroll
        for (int og = 0; og < 2; ++og) {
            v4i a2 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a0 = {0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
                if (kb < nkbh) {
                    const v4i A = *reinterpret_cast<const v4i*>(rect + (16 * mt + l15) * RSTR + goff[og] + kb * 64 + gq * 16);
                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, BH[og][kb][0], a2, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, BH[og][kb][1], a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, BH[og][kb][2], a0, 0, 0, 0);
                }
            *reinterpret_cast<unsigned*>(tt + (og * 16 + l15) * TSTR + 16 * mt + 4 * gq) = combine4(a2, a1, a0, hb[og], 0x80808080u);
        }
    }
    sync();

    // 3. vertical pass: 16 x 16 outputs per (row group, column group) pair; four waves take one pair each
    for (int combo = wave; combo < 4; combo += WAVES) {
        const int yt = combo >> 1, xt = combo & 1;
        if (WAVES == 1 && xt == 0) { gy = min(2 * ty + yt, ngv - 1); load_bv(); }
        const int voff = vws[gy] - r0, vb = vbias[gy * 16 + l15];
        v4i a2 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a0 = {0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            if (kb < nkbv) {
                const v4i A = *reinterpret_cast<const v4i*>(tt + (xt * 16 + l15) * TSTR + voff + kb * 64 + gq * 16);
                a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, BV[kb][0], a2, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, BV[kb][1], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, BV[kb][2], a0, 0, 0, 0);
            }
        const int y = 32 * ty + 16 * yt + l15, x = 32 * tx + 16 * xt + 4 * gq;
        if (y < S && x < S && 2 * ty + yt < ngv)
            *reinterpret_cast<unsigned*>(out + ((size_t)plane * S + y) * S + x) = combine4(a2, a1, a0, vb, 0u);
    }
}

}  // namespace wise

extern "C" int wise_preproc_u8(const wise_preproc_plan* plan, const void* dev_tables, const uint8_t* frames, int n,
                               uint8_t* out, void* stream) {
    WISE_CHECK_ARG(plan && dev_tables && frames && out, "preproc: null argument");
    WISE_CHECK_ARG(n >= 1 && n <= (1 << 20), "preproc: n=%d", n);
    HostPlan hp;
    int rc = build_plan(plan->H, plan->W, plan->S, plan->reserved & 1, hp);
    if (rc) return rc;
    const wise_preproc_plan& p = hp.p;
    WISE_CHECK_ARG(p.table_bytes == plan->table_bytes && p.tile == plan->tile && p.lds_bytes == plan->lds_bytes,
                   "preproc: plan was not made by wise_preproc_plan_init for %dx%d -> %d", p.H, p.W, p.S);
    WISE_CHECK_ARG(((uintptr_t)dev_tables & 15) == 0 && ((uintptr_t)out & 3) == 0, "preproc: tables/out misaligned");
    const int planes = n * 3;
    const int nt = (p.S + p.tile - 1) / p.tile;
    const long long blocks = (long long)((planes + 7) / 8) * 8 * nt * nt;
    WISE_CHECK_ARG(blocks < (1ll << 31), "preproc: too many tiles (%lld)", blocks);
    const int CS = p.max_cols4 | 1, RS = p.max_rows4 | 1;
    const bool aligned = (p.W % 4 == 0) && (((uintptr_t)frames & 3) == 0);
    hipStream_t st = (hipStream_t)stream;
    if (hp.mfma_ok && (plan->reserved & PREPROC_MFMA_FLAG)) {
        // the matrix-core form: one wave per 32 x 32 tile; its tables sit behind the dot-product kernel's in the blob
        const int nt32 = (p.S + 31) / 32;
        const long long mblocks = (long long)((planes + 7) / 8) * 8 * nt32 * nt32;
        WISE_CHECK_ARG(mblocks < (1ll << 31), "preproc: too many tiles (%lld)", mblocks);
        const size_t valu_bytes = (size_t)4 * ((size_t)2 * p.S + (size_t)p.S * (p.ndh + p.ndv) * 4 + (size_t)4 * nt);
        const int* mtab = reinterpret_cast<const int*>(static_cast<const unsigned char*>(dev_tables) + valu_bytes);
        const unsigned char* hb = reinterpret_cast<const unsigned char*>(mtab + hp.m_ints);
        const unsigned char* vb = hb + (size_t)hp.mh.ng * hp.mh.nkb * 3 * 1024;
        const int nkb = hp.mh.nkb > hp.mv.nkb ? hp.mh.nkb : hp.mv.nkb;
        const int tt_off = hp.mv.max_tn * hp.m_rstr;
        const bool four = (plan->reserved & PREPROC_MFMA4_FLAG) != 0;   // four waves per tile instead of one
#define WISE_RESIZE_MFMA(AL, NK)                                                                                             \
    do {                                                                                                                     \
        if (four)                                                                                                            \
            hipLaunchKernelGGL((clip_resize_mfma_kernel<AL, NK, 4>), dim3((unsigned)mblocks), dim3(256), hp.m_lds, st, frames, \
                               planes, p.H, p.W, p.S, hp.mh.nkb, hp.mv.nkb, hp.m_rstr, hp.m_tstr, tt_off, mtab, hp.mh.ng,     \
                               hp.mv.ng, hb, vb, out);                                                                       \
        else                                                                                                                 \
            hipLaunchKernelGGL((clip_resize_mfma_kernel<AL, NK, 1>), dim3((unsigned)mblocks), dim3(64), hp.m_lds, st, frames, \
                               planes, p.H, p.W, p.S, hp.mh.nkb, hp.mv.nkb, hp.m_rstr, hp.m_tstr, tt_off, mtab, hp.mh.ng,     \
                               hp.mv.ng, hb, vb, out);                                                                       \
    } while (0)
        if (aligned) { if (nkb == 1) WISE_RESIZE_MFMA(true, 1); else if (nkb == 2) WISE_RESIZE_MFMA(true, 2); else WISE_RESIZE_MFMA(true, 4); }
        else { if (nkb == 1) WISE_RESIZE_MFMA(false, 1); else if (nkb == 2) WISE_RESIZE_MFMA(false, 2); else WISE_RESIZE_MFMA(false, 4); }
#undef WISE_RESIZE_MFMA
        WISE_LAUNCH_CHECK("clip_resize_mfma_kernel");
        return WISE_OK;
    }
    if (aligned)
        hipLaunchKernelGGL(clip_resize_kernel<true>, dim3((unsigned)blocks), dim3(256), p.lds_bytes, st, frames, planes,
                           p.H, p.W, p.S, p.tile, p.ndh, p.ndv, CS, RS, p.max_rows4, (const int*)dev_tables, out);
    else
        hipLaunchKernelGGL(clip_resize_kernel<false>, dim3((unsigned)blocks), dim3(256), p.lds_bytes, st, frames, planes,
                           p.H, p.W, p.S, p.tile, p.ndh, p.ndv, CS, RS, p.max_rows4, (const int*)dev_tables, out);
    WISE_LAUNCH_CHECK("clip_resize_kernel");
    return WISE_OK;
}
