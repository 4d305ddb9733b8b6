// 1024-point complex FFT of one block of 256 threads in LDS (denoiser.hip: STFT and ISTFT of the bias denoiser; vocos.hip: the ISTFT head).
// Five radix-4 Stockham autosort passes over two buffers of 1024 complex: thread i of pass p (p = 1, 4, 16, 64, 256) takes a[i + 256 r],
// r = 0..3, multiplies by the twiddles exp(-2 pi i r k / (4 p)), k = i % p, and writes the radix-4 butterfly to b[4 (i - k) + k + p r]; the
// result is in natural order.  The indexing was checked against numpy.fft in double before the kernels were written (4e-14).  An inverse
// transform of a Hermitian spectrum X runs as Re(FFT(conj X)) / 1024.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <vector>

namespace ttsamd {

// twiddles exp(-2 pi i m / 1024) as (cos, -sin) pairs, rounded once from double; appended at an 8-byte-aligned offset, which is returned
inline int64_t fft1024_append_twiddles(std::vector<float>& blob) {
    if (blob.size() & 1) blob.push_back(0.f);
    const int64_t off = (int64_t)blob.size();
    const double two_pi = 6.283185307179586476925286766559;
    for (int m = 0; m < 1024; ++m) {
        blob.push_back((float)std::cos(two_pi * m / 1024));
        blob.push_back((float)(-std::sin(two_pi * m / 1024)));
    }
    return off;
}

#ifdef __HIPCC__
__device__ __forceinline__ float2 fft_cmul(const float2 a, const float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in: a (natural order); out: b (five passes: an odd number), natural order; both buffers are clobbered; tw = the 1024 twiddles (LDS).
// Ends with a __syncthreads(); the caller synchronises between filling `a` and the call.
__device__ __forceinline__ void fft1024_stockham(float2* a, float2* b, const float2* tw, const int i) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int p = 1 << (2 * s);
        const int k = i & (p - 1), j = ((i - k) << 2) + k, tstep = 256 >> (2 * s);
        float2 u0 = a[i], u1 = a[i + 256], u2 = a[i + 512], u3 = a[i + 768];
        if (s > 0) {
            u1 = fft_cmul(u1, tw[(k * tstep) & 1023]);
            u2 = fft_cmul(u2, tw[(2 * k * tstep) & 1023]);
            u3 = fft_cmul(u3, tw[(3 * k * tstep) & 1023]);
        }
        const float2 a0 = make_float2(u0.x + u2.x, u0.y + u2.y), a1 = make_float2(u0.x - u2.x, u0.y - u2.y);
        const float2 a2 = make_float2(u1.x + u3.x, u1.y + u3.y);
        const float2 a3 = make_float2(u1.y - u3.y, -(u1.x - u3.x));            // (u1 - u3) * (-i)
        b[j] = make_float2(a0.x + a2.x, a0.y + a2.y);
        b[j + p] = make_float2(a1.x + a3.x, a1.y + a3.y);
        b[j + 2 * p] = make_float2(a0.x - a2.x, a0.y - a2.y);
        b[j + 3 * p] = make_float2(a1.x - a3.x, a1.y - a3.y);
        __syncthreads();
        float2* t = a; a = b; b = t;
    }
}
#endif

}  // namespace ttsamd
