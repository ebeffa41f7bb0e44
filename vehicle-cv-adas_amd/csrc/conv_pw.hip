// conv_pw.hip -- pointwise (1x1) convolution, stride 1 or 2: the C2f/SPPF/Detect 1x1 layers of the YOLO graphs and the
// ResNet projection shortcuts.  These layers are HBM/L2-bound (arithmetic intensity = 2*Cin*Cout/(2*(Cin+Cout)) FLOP/B
// <= 128 for every YOLOv8n instance, ridge ~312), so the kernel is organised around streaming, not tiles:
//   * the whole weight matrix sits in LDS in MFMA-fragment order (one contiguous 1 KB ds_read_b128 per (feature tile,
//     K step), conflict-free by construction), loaded once per persistent workgroup;
//   * every wave owns 16 consecutive output pixels at a time and loads their activations straight from HBM into
//     MFMA B-operand registers (lane = (pixel, 8-channel group): 16 B per lane, 64 contiguous bytes per pixel per
//     K step) -- no LDS staging, no barrier in the pixel loop, latency hidden by 16-32 resident waves per CU;
//   * loop over feature tiles: KS MFMAs each, fused bias + SiLU/ReLU, 8 B (bf16) / 16 B (fp32) store per lane.
// Algorithmic bytes per launch: pixels * (Cin + Cout) * 2 B (+ Cout*Cin*2 B of weights, L2-resident).
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float pf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t pu32x4;

struct PwDev {
    const uint16_t* in;
    const uint16_t* wfrag;  // [NT][KS][64][8] bf16 fragment order (same packing as CONV_FC)
    const float* bias;
    void* out;
    int in_cs, in_coff, cin;
    int out_cs, out_coff, cout, out_f32;
    int M;                  // output pixels (all frames)
    int stride, Wo, HoWo, W, HW;  // stride-2 address mapping
    int act;
    int NT;                 // feature tiles (cout_pad16 / 16)
    int NTL;                // feature tiles one workgroup owns (blockIdx.y selects the range; NT when the weights fit LDS whole)
    int mtiles;
    int wide;               // 16-byte stores (two feature tiles per trip); ADAS_NO_PW_WIDE=1 clears it
    const uint16_t* up;     // half-resolution source of the first up_ks K steps (nearest-neighbour 2x upsample folded in), or null
    int up_cs, up_coff, up_ks, up_W, up_HW;
};

template <typename E, int KS, bool TAIL>
__global__ __launch_bounds__(512) void conv_pw_kernel(PwDev a) {
    E::enter();
    extern __shared__ __attribute__((aligned(16))) uint16_t wl[];  // [NTL][KS][64][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    // feature tiles [nt0, nt0 + ntl) belong to this workgroup: layers whose whole weight matrix exceeds the LDS budget are
    // split over blockIdx.y (the activations are then read once per range; on the small maps this happens they sit in L2)
    const int nt0 = blockIdx.y * a.NTL;
    const int ntl = a.NT - nt0 < a.NTL ? a.NT - nt0 : a.NTL;
    float* bl = reinterpret_cast<float*>(wl + (size_t)a.NTL * KS * 512);  // [NTL*16] bias, behind the weights
    {   // weights: contiguous copy, eight 16-byte loads in flight per thread (elem16.h stage_lds16); bias too (a global bias load inside the feature-tile loop
        // costs one exposed L2 round trip per tile: hipcc waits vmcnt(0) right behind it)
        const int n16 = ntl * KS * 64;
        const uint16_t* wsrc = a.wfrag + (size_t)nt0 * KS * 512;
        stage_lds16<512, 8>(wl, wsrc, n16, tid);
        for (int i = tid; i < ntl * 16; i += 512) bl[i] = a.bias[nt0 * 16 + i];   // bias is padded to a multiple of 128 entries
    }
    __syncthreads();
    constexpr bool WIDE_OK = KS <= 8;   // (the two-tile form doubles the weight fragments in flight; the widest instantiations keep the one-tile loop)
    const bool wide = !a.out_f32 && (((a.out_cs | a.out_coff) & 7) == 0) && a.wide;
    const int tail_valid = a.cin - (KS - 1) * 32;  // channels that exist in the last K step
    const bool tail_zero = TAIL && kg * 8 >= tail_valid;

    for (int mt = blockIdx.x * 8 + wave; mt < a.mtiles; mt += gridDim.x * 8) {
        const int m = mt * 16 + lrow;
        const bool ok = m < a.M;
        size_t ipix;
        if (a.stride == 1) {
            ipix = (size_t)(ok ? m : 0);
        } else {
            const int mm = ok ? m : 0;
            const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            ipix = (size_t)n * a.HW + (size_t)(oy * a.stride) * a.W + ox * a.stride;
        }
        const uint16_t* ip = a.in + ipix * a.in_cs + a.in_coff + kg * 8;
        pu32x4 xb[KS];
        if (a.up) {   // workgroup-uniform: the first up_ks K steps come from the half-resolution tensor (stride 1 only)
            const int mm = ok ? m : 0;
            const int n = mm / a.HW, rem = mm - n * a.HW;
            const int oy = rem / a.W, ox = rem - oy * a.W;
            const uint16_t* up = a.up + ((size_t)n * a.up_HW + (size_t)(oy >> 1) * a.up_W + (ox >> 1)) * a.up_cs + a.up_coff + kg * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                xb[ks] = ks < a.up_ks ? *reinterpret_cast<const pu32x4*>(up + ks * 32) : __builtin_nontemporal_load(reinterpret_cast<const pu32x4*>(ip + ks * 32));
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xb[ks] = __builtin_nontemporal_load(reinterpret_cast<const pu32x4*>(ip + ks * 32));
        }
        if (tail_zero) xb[KS - 1] = pu32x4{0u, 0u, 0u, 0u};

        const size_t obase = (size_t)(ok ? m : 0) * a.out_cs + a.out_coff + kg * 4 + nt0 * 16;
        int nt = 0;
        if (WIDE_OK && wide) {
            // two feature tiles per trip, 16-byte stores: v_permlane16_swap exchanges the odd 16-lane rows of tile nt with the even rows of
            // tile nt + 1, after which a lane owns 8 consecutive channels of its pixel (conv_halo.hip's epilogue): half the store
            // instructions, 64-byte instead of 32-byte runs per pixel
            auto actf = [&](float v) {
                if (a.act == ACT_SILU) return v * fast_rcp(1.0f + __expf(-v));
                if (a.act == ACT_RELU) return fmaxf(v, 0.f);
                if (a.act == ACT_LEAKY) return fmaxf(v, 0.1f * v);
                return v;
            };
            const size_t pbase = (size_t)(ok ? m : 0) * a.out_cs + a.out_coff + nt0 * 16;
            for (; nt + 1 < ntl; nt += 2) {
                pf32x4 acc0{0.f, 0.f, 0.f, 0.f}, acc1{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const pu32x4 w0 = *reinterpret_cast<const pu32x4*>(wl + ((size_t)(nt * KS + ks) * 64 + lane) * 8);
                    const pu32x4 w1 = *reinterpret_cast<const pu32x4*>(wl + ((size_t)((nt + 1) * KS + ks) * 64 + lane) * 8);
                    acc0 = E::mfma(w0, xb[ks], acc0);
                    acc1 = E::mfma(w1, xb[ks], acc1);
                }
                const float4 b0 = *reinterpret_cast<const float4*>(bl + nt * 16 + kg * 4), b1 = *reinterpret_cast<const float4*>(bl + (nt + 1) * 16 + kg * 4);
                const uint32_t x0 = E::pack2(actf(acc0[0] + b0.x), actf(acc0[1] + b0.y)), x1 = E::pack2(actf(acc0[2] + b0.z), actf(acc0[3] + b0.w));
                const uint32_t y0 = E::pack2(actf(acc1[0] + b1.x), actf(acc1[1] + b1.y)), y1 = E::pack2(actf(acc1[2] + b1.z), actf(acc1[3] + b1.w));
                const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const int c = (nt + (kg & 1)) * 16 + (kg >> 1) * 8;
                uint16_t* op = (uint16_t*)a.out + pbase + c;
                if (ok) {
                    if (nt0 * 16 + c + 8 <= a.cout) *reinterpret_cast<pu32x4*>(op) = pu32x4{s0[0], s1[0], s0[1], s1[1]};
                    else if (nt0 * 16 + c + 4 <= a.cout) *reinterpret_cast<uint2*>(op) = make_uint2(s0[0], s1[0]);
                }
            }
        }
        for (; nt < ntl; ++nt) {
            pf32x4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const pu32x4 wf = *reinterpret_cast<const pu32x4*>(wl + ((size_t)(nt * KS + ks) * 64 + lane) * 8);
                acc = E::mfma(wf, xb[ks], acc);
            }
            const int c = nt * 16 + kg * 4;
            if (!ok || nt0 * 16 + c >= a.cout) continue;
            const float4 b4 = *reinterpret_cast<const float4*>(bl + c);
            float v[4] = {acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w};
            if (a.act == ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * fast_rcp(1.0f + __expf(-v[r]));
            } else if (a.act == ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (a.act == ACT_LEAKY) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.1f * v[r]);
            }
            if (a.out_f32) {
                *reinterpret_cast<float4*>((float*)a.out + obase + nt * 16) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint2 q;
                q.x = E::pack2(v[0], v[1]);
                q.y = E::pack2(v[2], v[3]);
                *reinterpret_cast<uint2*>((uint16_t*)a.out + obase + nt * 16) = q;
            }
        }
    }
}

static const int PW_MAX_LDS = 150 * 1024;

// feature tiles per workgroup: all of them when the weights fit, otherwise an even split into at most 4 ranges
static int pw_tiles_per_wg(int nt, int ks) {
    for (int split = 1; split <= 4; ++split) {
        const int ntl = (nt + split - 1) / split;
        if ((size_t)ntl * ks * 1024 + (size_t)ntl * 64 <= (size_t)PW_MAX_LDS) return ntl;
    }
    return 0;
}

bool pw_applicable(int prec, int kh, int kw, int stride, int pad, int res_mode, const TView& in, const TView& out) {
    if (!prec_is16(prec) || in.f32) return false;
    if (kh != 1 || kw != 1 || pad != 0 || (stride != 1 && stride != 2) || res_mode != RES_NONE) return false;
    if (in.h == 1 && in.w == 1) return false;  // Linear layers have their own kernel
    if ((in.c & 7) || (in.cs & 7) || (in.coff & 7) || (out.c & 3) || (out.cs & 3) || (out.coff & 3)) return false;
    const int ks = (in.c + 31) / 32, nt = (out.c + 15) / 16;
    if (ks > 16) return false;
    if (!(ks <= 6 || ks == 8 || ks == 10 || ks == 12 || ks == 16)) return false;   // instantiated K-step counts (5 / 10: YOLOv8x's 160 / 320 channels)
    return pw_tiles_per_wg(nt, ks) > 0;
}

template <typename E, int KS>
static hipError_t pw_launch(const PwDev& d, bool tail, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_pw_kernel<E, KS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_MAX_LDS);
        (void)hipFuncSetAttribute((const void*)conv_pw_kernel<E, KS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_MAX_LDS);
        attr_done = true;
    }
    if (tail) hipLaunchKernelGGL((conv_pw_kernel<E, KS, true>), grid, dim3(512), lds, st, d);
    else hipLaunchKernelGGL((conv_pw_kernel<E, KS, false>), grid, dim3(512), lds, st, d);
    return hipGetLastError();
}
template <typename E>
static hipError_t pw_launch_ks(const PwDev& d, int ks, bool tail, dim3 grid, size_t lds, hipStream_t st) {
    switch (ks) {
        case 1: return pw_launch<E, 1>(d, tail, grid, lds, st);
        case 2: return pw_launch<E, 2>(d, tail, grid, lds, st);
        case 3: return pw_launch<E, 3>(d, tail, grid, lds, st);
        case 4: return pw_launch<E, 4>(d, tail, grid, lds, st);
        case 5: return pw_launch<E, 5>(d, tail, grid, lds, st);
        case 10: return pw_launch<E, 10>(d, tail, grid, lds, st);
        case 6: return pw_launch<E, 6>(d, tail, grid, lds, st);
        case 8: return pw_launch<E, 8>(d, tail, grid, lds, st);
        case 12: return pw_launch<E, 12>(d, tail, grid, lds, st);
        case 16: return pw_launch<E, 16>(d, tail, grid, lds, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv_pw(const ConvArgs& a, hipStream_t st) {
    PwDev d;
    d.in = (const uint16_t*)a.in.p; d.wfrag = (const uint16_t*)a.wgt; d.bias = a.bias; d.out = a.out.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c; d.out_f32 = a.out.f32;
    d.M = a.m; d.stride = a.stride; d.Wo = a.out.w; d.HoWo = a.out.h * a.out.w; d.W = a.in.w; d.HW = a.in.h * a.in.w;
    d.act = a.act;
    {
        const char* e = getenv("ADAS_NO_PW_WIDE");
        d.wide = (e && e[0] == '1') ? 0 : 1;
    }
    d.up = nullptr; d.up_cs = d.up_coff = d.up_ks = d.up_W = d.up_HW = 0;
    if (a.up_c > 0) {
        if (a.stride != 1 || (a.up_c & 31) || a.up.c != a.up_c || 2 * a.up.h != a.in.h || 2 * a.up.w != a.in.w || a.up.f32 || ((a.up.cs | a.up.coff) & 7)) return hipErrorInvalidValue;
        d.up = (const uint16_t*)a.up.p; d.up_cs = a.up.cs; d.up_coff = a.up.coff; d.up_ks = a.up_c / 32; d.up_W = a.up.w; d.up_HW = a.up.h * a.up.w;
    }
    const int ks = (a.in.c + 31) / 32;
    d.NT = (a.out.c + 15) / 16;
    d.mtiles = (a.m + 15) / 16;
    d.NTL = pw_tiles_per_wg(d.NT, ks);
    if (d.NTL <= 0) return hipErrorNotSupported;
    const int nsplit = (d.NT + d.NTL - 1) / d.NTL;
    const size_t lds = (size_t)d.NTL * ks * 1024 + (size_t)d.NTL * 64;
    // persistent grid: as many 8-wave workgroups as fit the LDS budget of 256 CUs, never more than the work
    int per_cu = (int)((160 * 1024) / (lds > 4096 ? lds : 4096));
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int gx = 256 * per_cu / nsplit;
    if (gx < 1) gx = 1;
    const int need = (d.mtiles + 7) / 8;
    if (gx > need) gx = need;
    const dim3 grid(gx, nsplit);
    const bool tail = (a.in.c & 31) != 0;
    if (a.prec == PREC_FP16) return pw_launch_ks<Fp16>(d, ks, tail, grid, lds, st);
    return pw_launch_ks<Bf16>(d, ks, tail, grid, lds, st);
}

}  // namespace adas
