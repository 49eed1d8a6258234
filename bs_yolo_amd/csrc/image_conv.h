// Shared pieces of the two kernels that convolve the input image (conv_first.hip, stem_fused.hip).
//
// The image arrives planar (BCHW, f16 or f32).  Both kernels park a patch of it in LDS as pixel-interleaved f16 with a
// zero fourth channel ([row][column][4], 8 bytes per pixel) and pack the 3x3 weights the same way: k = (kh, kw, c4),
// K = 36 padded to 48 = three 16-wide MFMA steps (host side: bs_yolo_amd/weights.py kind "first",
// bsy_conv_packed_dims with C1 == 3).  A lane's B fragment for one MFMA step is then two taps = two 8-byte LDS reads
// at compile-time offsets from its pixel's window origin; the planar layout it replaces cost sixteen 2-byte reads,
// their address arithmetic and the packing per step, which made both kernels VALU-bound.
#pragma once
#include "common.h"

#define IMGC_KROW 64  // halves per packed weight row (K = 36 -> round_up(36, 32))
#define IMGC_KSUB 3   // MFMA steps that hold non-zero weights

template <typename T>
struct ImgPix;  // 4 consecutive pixels of one image row, as loaded
template <>
struct ImgPix<half_t> {
    typedef half4 type;
    static __device__ __forceinline__ half4 zero() { return half4{0, 0, 0, 0}; }
    static __device__ __forceinline__ half4 cvt(half4 v) { return v; }
};
template <>
struct ImgPix<float> {
    typedef f32x4 type;
    static __device__ __forceinline__ f32x4 zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ half4 cvt(f32x4 v) { return half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}; }
};

// One work item of the patch load = 4 pixels x 3 channels of one image row.
template <typename T>
struct ImgItem {
    typename ImgPix<T>::type c[3];
};

// (iy, ix) = image coordinates of the item's first pixel; ix is a multiple of 4 and W % 4 == 0, so an item is either
// wholly inside the row or wholly outside (zero padding).
template <typename T>
__device__ __forceinline__ ImgItem<T> img_item_load(const T* __restrict__ img_n, int H, int W, int iy, int ix, bool active) {
    ImgItem<T> it;
    const bool ok = active && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        it.c[ch] = ImgPix<T>::zero();
        if (ok) it.c[ch] = *reinterpret_cast<const typename ImgPix<T>::type*>(img_n + ((size_t)ch * H + iy) * W + ix);
    }
    return it;
}

// Element-wise variant for images whose rows are not 4-pixel aligned (W % 4 != 0 or a misaligned base pointer).
template <typename T>
__device__ __forceinline__ ImgItem<T> img_item_load_slow(const T* __restrict__ img_n, int H, int W, int iy, int ix, bool active) {
    ImgItem<T> it;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        it.c[ch] = ImgPix<T>::zero();
        if (active && (unsigned)iy < (unsigned)H) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((unsigned)(ix + e) < (unsigned)W) it.c[ch][e] = img_n[((size_t)ch * H + iy) * W + ix + e];
        }
    }
    return it;
}

// dst = LDS address of the item's first pixel ([4 px][4 ch] f16 = 32 bytes, 16-byte aligned)
template <typename T>
__device__ __forceinline__ void img_item_park(const ImgItem<T>& it, half_t* dst) {
    const half4 r = ImgPix<T>::cvt(it.c[0]), g = ImgPix<T>::cvt(it.c[1]), b = ImgPix<T>::cvt(it.c[2]);
    const half_t z = (half_t)0.f;
    *reinterpret_cast<half8*>(dst) = half8{r[0], g[0], b[0], z, r[1], g[1], b[1], z};
    *reinterpret_cast<half8*>(dst + 8) = half8{r[2], g[2], b[2], z, r[3], g[3], b[3], z};
}

// B fragment of MFMA step s for the lane whose 3x3 window starts at LDS element `win` (f16 index of pixel (row, col),
// channel 0) in a patch of `ROWPX` pixels per row.  Lane half lh holds taps 4s + 2lh and 4s + 2lh + 1; taps 9..11
// carry zero weights, so they re-read tap 8 (finite data).
template <int ROWPX>
__device__ __forceinline__ half8 img_frag(const half_t* __restrict__ patch, int win, int s, int lh) {
    half8 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ta = 4 * s + h, tb = ta + 2;  // lh = 0 / 1
        const int ca = ta < 9 ? ta : 8, cb = tb < 9 ? tb : 8;
        const int oa = ((ca / 3) * ROWPX + ca % 3) * 4, ob = ((cb / 3) * ROWPX + cb % 3) * 4;
        const half4 v = *reinterpret_cast<const half4*>(patch + win + (lh ? ob : oa));
        f[4 * h + 0] = v[0]; f[4 * h + 1] = v[1]; f[4 * h + 2] = v[2]; f[4 * h + 3] = v[3];
    }
    return f;
}
