// Device-side helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

// Exact unsigned 32-bit division by a launch-invariant divisor
// (Granlund & Montgomery round-up method): q = n / d for every n < 2^32.
struct FastDiv {
    unsigned m, s1, s2, d;
    FastDiv() : m(1), s1(0), s2(0), d(1) {}
    explicit FastDiv(unsigned div) : d(div) {
        unsigned l = 0;
        while ((1ull << l) < div) ++l;  // l = ceil(log2 d)
        m = (unsigned)((((1ull << l) - div) << 32) / div) + 1;
        s1 = l < 1 ? l : 1;
        s2 = l > 0 ? l - 1 : 0;
    }
    __device__ __forceinline__ unsigned div(unsigned n) const {
        unsigned t = __umulhi(m, n);
        return (t + ((n - t) >> s1)) >> s2;
    }
    __device__ __forceinline__ void divmod(unsigned n, unsigned &q, unsigned &r) const {
        q = div(n);
        r = n - q * d;
    }
};

// The fused tail of a conv: what layer.BatchNorm / Add / ReLU / LeakyReLU
// (layer.py:125-127, 93-95, 44-51) would do to the conv output one by one.
// Multiply and add stay separate roundings like numpy's two passes.
struct Epilogue {
    const float *bias, *scale, *shift, *res;
    int act;
    float la, lb;  // leakyrelu: alpha, (1-alpha) rounded to fp32 (layer.py:49)
    int res_post;  // residual is added AFTER the activation (conv->bn->leakyrelu->add, YOLO-v3's blocks)
};

// act_code = PL_ACT_* optionally OR-ed with PL_ACT_RES_AFTER (16)
inline Epilogue make_epilogue(const float *bias, const float *scale, const float *shift, const float *res,
                              int act_code, double alpha) {
    return Epilogue{bias, scale, shift, res, act_code & 15, (float)alpha, (float)(1.0 - alpha), (act_code >> 4) & 1};
}

__device__ __forceinline__ float relu_ref(float v) {
    return v > 0.f ? v : __fmul_rn(v, 0.f);  // x*(x>0): negatives -> -0, NaN stays
}

__device__ __forceinline__ float leaky_ref(float v, float a, float b) {
    return __fmul_rn(__fadd_rn(v > 0.f ? b : 0.f, a), v);
}

__device__ __forceinline__ float apply_epilogue(const Epilogue &e, float v, int c, size_t idx) {
    if (e.bias) v = __fadd_rn(v, e.bias[c]);
    if (e.scale) v = __fmul_rn(v, e.scale[c]);
    if (e.shift) v = __fadd_rn(v, e.shift[c]);
    if (e.res && !e.res_post) v = __fadd_rn(v, e.res[idx]);
    if (e.act == 1) v = relu_ref(v);
    else if (e.act == 2) v = leaky_ref(v, e.la, e.lb);
    if (e.res && e.res_post) v = __fadd_rn(v, e.res[idx]);
    return v;
}
