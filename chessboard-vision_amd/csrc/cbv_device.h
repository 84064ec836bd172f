// Device-side helpers shared by the kernels.  gfx950 only (wave = 64 lanes).
#pragma once
#include "cbv_internal.h"

#define WAVE 64

__device__ __forceinline__ int d_round_f(float v) { return __float2int_rn(v); }  // cvRound: half to even
__device__ __forceinline__ int d_floor_f(float v) { return __float2int_rd(v); }  // cvFloor
__device__ __forceinline__ int d_round_d(double v) { return __double2int_rn(v); }
__device__ __forceinline__ u8 d_sat8(int v) { return (u8)min(max(v, 0), 255); }
__device__ __forceinline__ u8 d_sat8_f(float v) { return d_sat8(d_round_f(v)); }

// Correctly rounded float32 square root (np.sqrt on float32), spelled out: v_sqrt_f32 is good to 1 ulp, the two residuals
// decide between the result and its neighbours.  Not left to `__fsqrt_rn`: hipcc expands that to this very sequence in
// one kernel and to the bare instruction in another (k_squares_ema, where a quarter of the results were 1 ulp off).
// Zero, infinity and NaN pass through (every comparison with the NaN residuals is false); x is a normal number otherwise.
__device__ __forceinline__ float d_sqrt_rn(float x)
{
    float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __int_as_float(__float_as_int(y) - 1), yp = __int_as_float(__float_as_int(y) + 1);
    const float rm = __fmaf_rn(-ym, y, x), rp = __fmaf_rn(-yp, y, x);
    if (rm <= 0.f) y = ym;
    if (rp > 0.f) y = yp;
    return y;
}

// 1.0f / x, correctly rounded, for x in [1, 64): v_rcp_f32 (1 ulp) and one fused Newton step.  Checked EXHAUSTIVELY, every
// float of [1, 128) (one binade of margin), against the IEEE division of the host and of the device
// (tools/probe_rcp_exact.hip: 0 of 58 720 256 differ).  Three instructions where hipcc's expansion of `/` takes ten.  Callers guarantee the interval.
__device__ __forceinline__ float d_rcp_1_64(float x)
{
    const float y = __builtin_amdgcn_rcpf(x);
    return __fmaf_rn(__fmaf_rn(-x, y, 1.0f), y, y);
}

// cv2.normalize(NORM_MINMAX, 0, 255) as a byte map: entry t for a frame whose bytes span [vmin, vmax]: scale and shift in
// double, one float fma per value (cvt_32f), round half to even, saturate
__device__ __forceinline__ u8 d_norm_lut_entry(int vmin, int vmax, int t)
{
    const double smin = (double)vmin, smax = (double)vmax;
    const double scale = 255.0 * (smax - smin > 2.2204460492503131e-16 ? 1. / (smax - smin) : 0.);
    const double shift = 0.0 - smin * scale;
    return d_sat8_f(__fmaf_rn((float)t, (float)scale, (float)shift));
}

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int d_reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

#define D_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

// cooperative copy of a table into LDS (n bytes, both 4-byte aligned, n % 4 == 0)
__device__ __forceinline__ void lds_copy(void* dst, const void* src, int nbytes)
{
    const u32* s = (const u32*)src;
    u32* d = (u32*)dst;
    for (int i = threadIdx.x; i < nbytes / 4; i += blockDim.x) d[i] = s[i];
}

// wave-level reductions (64 lanes)
__device__ __forceinline__ u32 wave_sum_u32(u32 v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}

// XCD-aware tile order: hardware deals consecutive workgroup ids round-robin
// over the 8 XCDs; remap so that each XCD walks one contiguous range of tiles
// (neighbouring tiles then share halo rows through one L2).  Bijective for
// any n (MI355X guide, T1).
__device__ __forceinline__ int xcd_remap(int bid, int n)
{
    int xcd = bid & 7, q = n >> 3, r = n & 7;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// BGR2GRAY 8-bit (15-bit coefficients)
__device__ __forceinline__ int d_gray(int b, int g, int r) { return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15; }

// four BGR pixels (12 bytes) held in three registers
struct Px4 {
    u32 d[3];
};
__device__ __forceinline__ int px_get(const Px4& p, int i) { return (int)((p.d[i >> 2] >> ((i & 3) * 8)) & 255u); }
__device__ __forceinline__ void px_set(Px4& p, int i, int v) { p.d[i >> 2] |= ((u32)v & 255u) << ((i & 3) * 8); }
