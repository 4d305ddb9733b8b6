// Micro-benchmark of Tacotron2 LSTMCell GEMV variants (tools only; the winner lives in csrc/tacotron2.hip).
// Alternates the attention-rnn (K = 256+640+1024) and decoder-rnn (K = 1024+640+1024) problems like a decoder
// step does, so cache state matches production.   hipcc --offload-arch=gfx950 -O3 -o bin/gemv_bench gemv_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
constexpr int BC = 8;

// ---------------------------------------------------------------- v1: block per unit, K over threads, x from L2
template <int NR>
__device__ __forceinline__ void block_dots_acc(const float* __restrict__ w, int64_t row_stride, int K,
                                               const float* __restrict__ x, int x_stride, int b0, int B,
                                               float (&acc)[NR][BC]) {
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        float4 wv[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) wv[r] = *reinterpret_cast<const float4*>(w + r * row_stride + k);
#pragma unroll
        for (int bb = 0; bb < BC; ++bb) {
            const int b = min(b0 + bb, B - 1);
            const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)b * x_stride + k);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                acc[r][bb] = fmaf(wv[r].x, xv.x, fmaf(wv[r].y, xv.y, fmaf(wv[r].z, xv.z, fmaf(wv[r].w, xv.w, acc[r][bb]))));
        }
    }
}
template <int NR>
__device__ __forceinline__ void block_dots_reduce(float (&acc)[NR][BC], float (*red)[BC], float (*part)[NR][BC]) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int bb = 0; bb < BC; ++bb) {
            float v = acc[r][bb];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) part[wid][r][bb] = v;
        }
    __syncthreads();
    if (threadIdx.x < NR * BC) {
        const int r = threadIdx.x / BC, bb = threadIdx.x % BC;
        red[r][bb] = part[0][r][bb] + part[1][r][bb] + part[2][r][bb] + part[3][r][bb];
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void lstm_v1(const float* __restrict__ x1, int n1, const float* __restrict__ x2, int n2,
                                               const float* __restrict__ h_in, float* __restrict__ c,
                                               const float* __restrict__ wih, const float* __restrict__ whh,
                                               const float* __restrict__ bias, float* __restrict__ h_out, int B, int H) {
    __shared__ float red[4][BC], part[4][4][BC];
    const int u = blockIdx.x;
    const int K1 = n1 + n2;
    for (int b0 = 0; b0 < B; b0 += BC) {
        float acc[4][BC];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int bb = 0; bb < BC; ++bb) acc[g][bb] = 0.f;
        block_dots_acc<4>(wih + (int64_t)u * K1, (int64_t)H * K1, n1, x1, n1, b0, B, acc);
        block_dots_acc<4>(wih + (int64_t)u * K1 + n1, (int64_t)H * K1, n2, x2, n2, b0, B, acc);
        block_dots_acc<4>(whh + (int64_t)u * H, (int64_t)H * H, H, h_in, H, b0, B, acc);
        block_dots_reduce<4>(acc, red, part);
        if (threadIdx.x < BC && b0 + threadIdx.x < B) {
            const int bb = threadIdx.x, b = b0 + bb;
            const float gi = red[0][bb] + bias[u], gf = red[1][bb] + bias[H + u];
            const float gg = red[2][bb] + bias[2 * H + u], go = red[3][bb] + bias[3 * H + u];
            const float cn = sigmoidf_(gf) * c[(int64_t)b * H + u] + sigmoidf_(gi) * tanhf(gg);
            c[(int64_t)b * H + u] = cn;
            h_out[(int64_t)b * H + u] = sigmoidf_(go) * tanhf(cn);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- v4: column-parallel, x in registers
// thread = one float4 column group of [x1|x2|h]; block = UPB hidden units, processed RP rows per pass
template <int UPB, int NW>
__global__ __launch_bounds__(NW * 64) void lstm_v4(const float* __restrict__ x1, int n1, const float* __restrict__ x2, int n2,
                                                   const float* __restrict__ h_in, float* __restrict__ c,
                                                   const float* __restrict__ wih, const float* __restrict__ whh,
                                                   const float* __restrict__ bias, float* __restrict__ h_out, int B, int H) {
    __shared__ float part[NW][64], gates[UPB * 4 * BC];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K1 = n1 + n2, Kt = K1 + H, K4 = Kt / 4;
    const bool act = tid < K4;
    const int k = 4 * min(tid, K4 - 1);
    const int u0 = blockIdx.x * UPB;
    for (int b0 = 0; b0 < B; b0 += BC) {
        float4 xv[BC];
#pragma unroll
        for (int bb = 0; bb < BC; ++bb) {
            const int b = min(b0 + bb, B - 1);
            const float* src = k < n1 ? x1 + (int64_t)b * n1 + k
                             : k < K1 ? x2 + (int64_t)b * n2 + (k - n1) : h_in + (int64_t)b * H + (k - K1);
            xv[bb] = *reinterpret_cast<const float4*>(src);
            if (!act) xv[bb] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int p = 0; p < UPB / 2; ++p) {           // pass = 2 units = 8 gate rows
            float4 w[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int64_t row = (int64_t)(r & 3) * H + (u0 + 2 * p + (r >> 2));
                w[r] = *reinterpret_cast<const float4*>(k < K1 ? wih + row * K1 + k : whh + row * H + (k - K1));
            }
            float v[64];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int bb = 0; bb < BC; ++bb)
                    v[r * 8 + bb] = fmaf(w[r].x, xv[bb].x, fmaf(w[r].y, xv[bb].y, fmaf(w[r].z, xv[bb].z, w[r].w * xv[bb].w)));
#pragma unroll
            for (int s = 0; s < 6; ++s) {             // transposing butterfly: lane ends with index == lane
                const int m = 32 >> s, half = 32 >> s;
                const bool upper = (lane & m) != 0;
#pragma unroll
                for (int j = 0; j < half; ++j) {
                    const float send = upper ? v[j] : v[j + half];
                    const float keep = upper ? v[j + half] : v[j];
                    v[j] = keep + __shfl_xor(send, m);
                }
            }
            part[wid][lane] = v[0];
            __syncthreads();
            if (tid < 64) {
                float g = 0.f;
#pragma unroll
                for (int q = 0; q < NW; ++q) g += part[q][tid];
                gates[p * 64 + tid] = g;      // [unit-in-pass (2)][gate (4)][b (8)]
            }
            __syncthreads();
        }
        if (tid < UPB * BC) {
            const int uu = tid / BC, bb = tid % BC, b = b0 + bb, u = u0 + uu;
            if (b < B) {
                const float* gp = gates + (uu >> 1) * 64 + (uu & 1) * 32 + bb;
                const float gi = gp[0] + bias[u], gf = gp[8] + bias[H + u], gg = gp[16] + bias[2 * H + u], go = gp[24] + bias[3 * H + u];
                const float cn = sigmoidf_(gf) * c[(int64_t)b * H + u] + sigmoidf_(gi) * tanhf(gg);
                c[(int64_t)b * H + u] = cn;
                h_out[(int64_t)b * H + u] = sigmoidf_(go) * tanhf(cn);
            }
        }
        __syncthreads();
    }
}

struct Prob {
    int n1, n2, H, B;
    float *x1, *x2, *h, *c, *wih, *whh, *bias, *hout;
    std::vector<float> hx1, hx2, hh, hc, hwih, hwhh, hb;
};
static float frand() { return (rand() / (float)RAND_MAX) * 2.f - 1.f; }
static void make(Prob& p, int n1, int n2, int H, int B) {
    p.n1 = n1; p.n2 = n2; p.H = H; p.B = B;
    const int K1 = n1 + n2;
    p.hx1.resize((size_t)B * n1); p.hx2.resize((size_t)B * n2); p.hh.resize((size_t)B * H); p.hc.resize((size_t)B * H);
    p.hwih.resize((size_t)4 * H * K1); p.hwhh.resize((size_t)4 * H * H); p.hb.resize(4 * H);
    for (auto* v : {&p.hx1, &p.hx2, &p.hh, &p.hc}) for (auto& e : *v) e = frand();
    for (auto& e : p.hwih) e = frand() * 0.03f;
    for (auto& e : p.hwhh) e = frand() * 0.03f;
    for (auto& e : p.hb) e = frand() * 0.1f;
    auto up = [](float** d, const std::vector<float>& h) { hipMalloc((void**)d, h.size() * 4); hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice); };
    up(&p.x1, p.hx1); up(&p.x2, p.hx2); up(&p.h, p.hh); up(&p.c, p.hc); up(&p.wih, p.hwih); up(&p.whh, p.hwhh); up(&p.bias, p.hb);
    hipMalloc((void**)&p.hout, (size_t)B * H * 4);
}
static double check(Prob& p) {   // compares h_out after ONE application on fresh c
    const int K1 = p.n1 + p.n2, H = p.H;
    std::vector<float> out((size_t)p.B * H);
    hipMemcpy(out.data(), p.hout, out.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int b = 0; b < p.B; ++b)
        for (int u = 0; u < H; u += 37) {
            double g[4];
            for (int q = 0; q < 4; ++q) {
                double a = p.hb[q * H + u];
                const float* wr = &p.hwih[((size_t)q * H + u) * K1];
                for (int k = 0; k < p.n1; ++k) a += (double)wr[k] * p.hx1[(size_t)b * p.n1 + k];
                for (int k = 0; k < p.n2; ++k) a += (double)wr[p.n1 + k] * p.hx2[(size_t)b * p.n2 + k];
                const float* hr = &p.hwhh[((size_t)q * H + u) * H];
                for (int k = 0; k < H; ++k) a += (double)hr[k] * p.hh[(size_t)b * H + k];
                g[q] = a;
            }
            auto sg = [](double x) { return 1.0 / (1.0 + std::exp(-x)); };
            const double cn = sg(g[1]) * p.hc[(size_t)b * H + u] + sg(g[0]) * std::tanh(g[2]);
            const double hn = sg(g[3]) * std::tanh(cn);
            worst = std::max(worst, std::fabs(hn - out[(size_t)b * H + u]));
        }
    return worst;
}

template <typename F>
static void bench(const char* name, Prob& a, Prob& d, F launch) {
    hipMemcpy(a.c, a.hc.data(), a.hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d.c, d.hc.data(), d.hc.size() * 4, hipMemcpyHostToDevice);
    launch(a); launch(d);
    hipDeviceSynchronize();
    const double ea = check(a), ed = check(d);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) { launch(a); launch(d); }
    hipEventRecord(e0);
    const int it = 300;
    for (int i = 0; i < it; ++i) { launch(a); launch(d); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s B=%2d  %.2f us per (att+dec) pair   max err %.1e %.1e\n", name, a.B, ms * 1e3 / it, ea, ed);
}

int main() {
    for (int B : {8, 1, 32}) {
        Prob a, d;
        make(a, 256, 640, 1024, B);
        make(d, 1024, 640, 1024, B);
        bench("v1 block/unit, x from L2", a, d, [](Prob& p) {
            hipLaunchKernelGGL(lstm_v1, dim3(p.H), dim3(256), 0, 0, p.x1, p.n1, p.x2, p.n2, p.h, p.c, p.wih, p.whh, p.bias, p.hout, p.B, p.H); });
        bench("v4 column-parallel UPB=4", a, d, [](Prob& p) {
            const int K4 = (p.n1 + p.n2 + p.H) / 4;
            if (K4 <= 512) hipLaunchKernelGGL((lstm_v4<4, 8>), dim3(p.H / 4), dim3(512), 0, 0, p.x1, p.n1, p.x2, p.n2, p.h, p.c, p.wih, p.whh, p.bias, p.hout, p.B, p.H);
            else hipLaunchKernelGGL((lstm_v4<4, 11>), dim3(p.H / 4), dim3(704), 0, 0, p.x1, p.n1, p.x2, p.n2, p.h, p.c, p.wih, p.whh, p.bias, p.hout, p.B, p.H); });
        bench("v4 column-parallel UPB=2", a, d, [](Prob& p) {
            const int K4 = (p.n1 + p.n2 + p.H) / 4;
            if (K4 <= 512) hipLaunchKernelGGL((lstm_v4<2, 8>), dim3(p.H / 2), dim3(512), 0, 0, p.x1, p.n1, p.x2, p.n2, p.h, p.c, p.wih, p.whh, p.bias, p.hout, p.B, p.H);
            else hipLaunchKernelGGL((lstm_v4<2, 11>), dim3(p.H / 2), dim3(704), 0, 0, p.x1, p.n1, p.x2, p.n2, p.h, p.c, p.wih, p.whh, p.bias, p.hout, p.B, p.H); });
    }
    return 0;
}
