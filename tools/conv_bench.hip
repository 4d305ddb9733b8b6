// Standalone micro-benchmark of the MFMA conv engine (kernel experiments; not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DTTS_...] tools/conv_bench.hip -o /tmp/conv_bench
#define TTS_WITH_DIRECT 1
#include "../tts-arabic-pytorch_amd/csrc/common.hpp"
namespace ttsamd { bool direct_supported(const ConvParams& p); int32_t launch_direct(const ConvParams& p, hipStream_t stream); }
#include "../tts-arabic-pytorch_amd/csrc/conv_mfma.hip"
#include "conv_direct_f32.hip"
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
namespace ttsamd {
static thread_local std::string g_err;
int32_t launch_conv_bf16_any(const ConvParams&, hipStream_t) { return -1; }   // fp32-only tool build
void set_error(const char* fmt, ...) { char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); g_err = buf; fprintf(stderr, "ERR %s\n", buf); }
}
using namespace ttsamd;
int main(int argc, char** argv) {
    struct Shape { int B, cin, cout, k, dil, L; };
    std::vector<Shape> shapes = {{8, 1024, 128, 3, 1, 28672}, {8, 1024, 128, 11, 5, 28672}, {32, 128, 128, 3, 1, 28672},
                                 {32, 128, 128, 7, 3, 28672}, {32, 64, 64, 3, 1, 57344}, {32, 32, 32, 3, 1, 114688}, {32, 32, 32, 11, 5, 114688}};
    if (getenv("PROD")) shapes = {{32, 256, 256, 3, 1, 3584}, {32, 256, 256, 7, 3, 3584}, {32, 256, 256, 11, 5, 3584},
                                  {32, 128, 128, 3, 1, 28672}, {32, 128, 128, 7, 3, 28672}, {32, 128, 128, 11, 5, 28672},
                                  {32, 64, 64, 3, 1, 57344}, {32, 64, 64, 7, 3, 57344}, {32, 64, 64, 11, 5, 57344},
                                  {32, 32, 32, 3, 1, 114688}, {32, 32, 32, 7, 3, 114688}, {32, 32, 32, 11, 5, 114688},
                                  {32, 384, 1536, 3, 1, 496}, {32, 1536, 384, 3, 1, 496}};
    if (const char* cs = getenv("CUSTOM")) {   // CUSTOM="B,C,k,dil,L;B,C,k,dil,L;..."
        shapes.clear();
        int B, C, k, d, L, n = 0;
        while (sscanf(cs, "%d,%d,%d,%d,%d%n", &B, &C, &k, &d, &L, &n) == 5) {
            shapes.push_back({B, C, C, k, d, L});
            cs += n;
            if (*cs == ';') ++cs;
        }
    }
    for (auto s : shapes) {
        const int cp = cout_padded(s.cout);
        float *x, *w, *y, *b;
        size_t nx = (size_t)s.B * s.cin * s.L, ny = (size_t)s.B * s.cout * s.L, nw = (size_t)s.cin * s.k * cp;
        hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&w, nw * 4); hipMalloc(&b, cp * 4);
        std::vector<float> hx(1 << 20), hw(nw);
        const bool zero = getenv("ZERO") != nullptr;
        for (auto& v : hx) v = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f;
        for (auto& v : hw) v = zero ? 0.f : ((float)rand() / RAND_MAX - 0.5f) * 0.05f;
        for (size_t o = 0; o < nx; o += hx.size()) hipMemcpy(x + o, hx.data(), std::min(hx.size(), nx - o) * 4, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice); hipMemset(b, 0, cp * 4);
        ConvParams p; std::memset(&p, 0, sizeof p);
        p.x = x; p.x_bs = (int64_t)s.cin * s.L; p.x_cs = s.L; p.w = w; p.bias = b;
        p.y = y; p.y_bs = (int64_t)s.cout * s.L; p.y_cs = s.L; p.y_ts = 1;
        if (s.cin >= s.cout) { p.res = x; p.r_bs = p.x_bs; p.r_cs = s.L; }   // residual read like c2
        float* rsep = nullptr;
        if (getenv("RES_SEP") && s.cin == s.cout) {   // production: the residual is another tensor than the conv input
            hipMalloc(&rsep, nx * 4); hipMemcpy(rsep, x, nx * 4, hipMemcpyDeviceToDevice); p.res = rsep;
        }
        if (getenv("NO_RES")) p.res = nullptr;        // c1 convs
        p.len_in_mul = p.len_out_mul = 1; p.Lin = p.Nout = s.L; p.Cin = s.cin; p.Cout = s.cout; p.CoutP = cp; p.K = s.k;
        p.dil = s.dil; p.pad = (s.k * s.dil - s.dil) / 2; p.n_phase = 1; p.in_slope = 0.1f; p.div = 1.f; p.batch = s.B;
        // RAGGED=<samples per frame>: per-utterance lengths like the bench workload (sum of 64 durations in [2,12]
        // frames, the longest = L / mul), so tiles past an utterance's end exit early as in production
        double valid_frac = 1.0;
        if (const char* rg = getenv("RAGGED")) {
            const int mul = strcmp(rg, "auto") == 0 ? std::max(1, s.L / 448) : atoi(rg), T = s.L / mul;
            std::vector<int64_t> hl(s.B);
            unsigned st = 12345u;
            int64_t mx = 0, sum = 0;
            for (auto& v : hl) { int t = 0; for (int i = 0; i < 64; ++i) { st = st * 1664525u + 1013904223u; t += 2 + (int)((st >> 16) % 11u); } v = t; mx = std::max<int64_t>(mx, t); }
            for (auto& v : hl) { v = getenv("RAGGED_FRAC") ? (int64_t)(T * atof(getenv("RAGGED_FRAC"))) : std::min<int64_t>(T, v * T / mx); sum += v; }   // RAGGED_FRAC=f: every utterance f x the padded length
            int64_t* dl; hipMalloc(&dl, s.B * 8); hipMemcpy(dl, hl.data(), s.B * 8, hipMemcpyHostToDevice);
            p.lens_in = dl; p.lens_out = dl; p.len_in_mul = p.len_out_mul = mul;
            valid_frac = (double)sum / ((double)T * s.B);
        }
#ifdef TTS_TIMING
        unsigned long long* tbuf = nullptr;
        const size_t tblocks = (size_t)((s.L + 31) / 32) * (cp / 32) * s.B;       // upper bound on blocks
        hipMalloc(&tbuf, tblocks * 8 * sizeof(unsigned long long));
        hipMemset(tbuf, 0, tblocks * 8 * sizeof(unsigned long long));
        p.timing = tbuf;
#endif
        auto do_launch = [&] { launch_conv(p, 0); };
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        // warm the clocks: the GPU drops to its idle state during the host-side setup above and needs tens of ms of
        // work to return to its compute clock (a cold 0.5 ms kernel ran at 1.7 GHz instead of 2.2-2.3, tools/conv_timeline.py)
        const int warm_ms = getenv("WARM_MS") ? atoi(getenv("WARM_MS")) : 150;
        {
            hipEvent_t w0, w1; hipEventCreate(&w0); hipEventCreate(&w1);
            hipEventRecord(w0, 0);
            float el = 0.f;
            int it = 0;
            do {
                for (int i = 0; i < 8; ++i) do_launch();
                hipEventRecord(w1, 0); hipEventSynchronize(w1);
                hipEventElapsedTime(&el, w0, w1);
            } while (el < (float)warm_ms && ++it < 10000);
        }
        hipEventRecord(e0, 0);
        const int n = getenv("ITERS") ? atoi(getenv("ITERS")) : 20;
        for (int i = 0; i < n; ++i) do_launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
        double fl = 2.0 * s.cout * s.cin * s.k * (double)s.B * s.L * valid_frac;
        printf("B%d cin%d cout%d k%d d%d L%d: %.3f ms %.1f TF (valid %.3f)\n", s.B, s.cin, s.cout, s.k, s.dil, s.L, ms, fl / ms / 1e9, valid_frac);
        fflush(stdout);
#ifdef TTS_TIMING
        {
            hipMemset(tbuf, 0, tblocks * 8 * sizeof(unsigned long long));
            for (int i = 0; i < 20; ++i) do_launch();      // the stamped launch is the last, clocks warm
            hipDeviceSynchronize();
            std::vector<unsigned long long> ht(tblocks * 8);
            hipMemcpy(ht.data(), tbuf, ht.size() * 8, hipMemcpyDeviceToHost);
            char fn[256]; snprintf(fn, sizeof fn, "gpurun_out/timing_c%d_k%d.csv", s.cin, s.k);
            FILE* f = fopen(fn, "w");
            if (f) {
                fprintf(f, "block,start,pro,main,end,hwid,xcc,e1,clk\n");
                for (size_t i = 0; i < tblocks; ++i)
                    if (ht[i * 8]) fprintf(f, "%zu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", i, ht[i * 8], ht[i * 8 + 1], ht[i * 8 + 2], ht[i * 8 + 3], ht[i * 8 + 4], ht[i * 8 + 5], ht[i * 8 + 6], ht[i * 8 + 7]);
                fclose(f);
            }
            hipFree(tbuf);
        }
#endif
        hipFree(x); hipFree(y); hipFree(w); hipFree(b); if (rsep) hipFree(rsep);
    }
    return 0;
}
