// Standalone micro-benchmark of the fused bf16 ResBlock pair (kernel experiments; not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DBFO_TIMING] tools/bfo_pair_bench.hip -o tools/bin/bfo_pair_bench
// Run:   bfo_pair_bench [C k dil L B]   (default: the nine production shapes at B = 32 x 448 frames)
// -DBFO_TIMING: per-block shader-clock stamps (start | window staged | phase A issued | exchanged | phase B issued | end).
#include "../tts-arabic-pytorch_amd/csrc/bfo_pair.hip"
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>
namespace ttsamd {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) { char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); g_err = buf; fprintf(stderr, "ERR %s\n", buf); }
void conv_log(const char*, int, int, int, int, int, int, int, int, int, int) {}
bool compact_order(const void* lens, int batch) { return lens != nullptr && batch > 1; }
int64_t bfo_packed_conv_elems(int cout, int cin, int k) { return (int64_t)((cin + 15) / 16) * k * 2 * ((cout + 31) & ~31) * 8; }
}
using namespace ttsamd;
int main(int argc, char** argv) {
    struct Shape { int C, k, dil, L, B; };
    std::vector<Shape> shapes;
    if (argc >= 6) shapes.push_back({atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5])});
    else
        for (int C : {128, 64, 32})
            for (auto kd : {std::pair<int, int>{3, 1}, {7, 3}, {11, 5}}) shapes.push_back({C, kd.first, kd.second, 448 * 8192 / C, 32});
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 20;
    for (auto s : shapes) {
        const size_t n16 = (size_t)s.B * (s.C / 8) * s.L;               // entries
        void *x, *y;
        hipMalloc(&x, n16 * 16); hipMalloc(&y, n16 * 16);
        std::vector<uint16_t> hx(1 << 20);
        for (auto& v : hx) { float f = ((float)rand() / RAND_MAX - 0.5f) * 2.f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
        for (size_t o = 0; o < n16 * 8; o += hx.size()) hipMemcpy((uint16_t*)x + o, hx.data(), std::min(hx.size(), n16 * 8 - o) * 2, hipMemcpyHostToDevice);
        const int64_t nw = bfo_packed_conv_elems(s.C, s.C, s.k);
        std::vector<uint16_t> hw(nw);
        for (auto& v : hw) { float f = ((float)rand() / RAND_MAX - 0.5f) * 0.1f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
        uint16_t *w1, *w2; float* b;
        hipMalloc(&w1, nw * 2); hipMalloc(&w2, nw * 2); hipMalloc(&b, 4 * s.C); hipMemset(b, 0, 4 * s.C);
        hipMemcpy(w1, hw.data(), nw * 2, hipMemcpyHostToDevice); hipMemcpy(w2, hw.data(), nw * 2, hipMemcpyHostToDevice);
        BfoPairParams p; memset(&p, 0, sizeof p);
        p.x = x; p.y = y; p.w1 = w1; p.w2 = w2; p.b1 = b; p.b2 = b; p.len_mul = 1; p.L = s.L; p.dil = s.dil; p.batch = s.B;
        p.mode = 0; p.div = 1.f; p.in_slope = 0.1f; p.mid_slope = 0.1f; p.out_slope = 0.1f;
        const int ts8 = (s.C == 128 ? 1 : s.C == 64 ? 2 : 4) * 256 - (s.k - 1);
        const bool small_ = (s.k == 3 && s.C <= 64) || (size_t)((s.L + ts8 - 1) / ts8) * s.B < 384;
        const int ts = (s.C == 128 ? 1 : s.C == 64 ? 2 : 4) * (small_ ? 128 : 256) - (s.k - 1);
        const size_t nblk = (size_t)((s.L + ts - 1) / ts) * s.B;
#ifdef BFO_TIMING
        unsigned long long* tim; hipMalloc(&tim, nblk * 128); hipMemset(tim, 0, nblk * 128); p.timing = tim;
#endif
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        // warm the clock
        for (int i = 0; i < 300; ++i) bfo_launch_pair(s.C, s.k, p, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) bfo_launch_pair(s.C, s.k, p, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / iters * 1e3, fl = 2.0 * 2.0 * s.C * s.C * s.k * (double)s.L * s.B, by = 2.0 * 2.0 * s.C * (double)s.L * s.B;
        printf("pair C=%d k=%d d=%d L=%d B=%d: %.1f us  %.0f TF (%.3f)  %.0f GB/s (%.3f)  blocks %zu\n", s.C, s.k, s.dil, s.L, s.B, us,
               fl / us / 1e6, fl / us / 1e6 / 2500.0, by / us / 1e3, by / us / 1e3 / 8000.0, nblk);
#ifdef BFO_TIMING
        std::vector<unsigned long long> ht(nblk * 16);
        hipMemcpy(ht.data(), tim, nblk * 128, hipMemcpyDeviceToHost);
        // stamp order in the kernel: 0 start | 1 staged | 2 A issued | 6 residual loads issued | 7 barrier 1 | 8 T written | 3 barrier 2 |
        //                            9 acc2 initialised | 4 B issued | 5 end
        const int order[10] = {0, 1, 2, 6, 7, 8, 3, 9, 4, 5};
        const char* names[9] = {"stage", "phase A", "res loads", "barrier 1", "T -> LDS", "barrier 2", "acc2 init", "phase B", "epilogue"};
        double d[9] = {0}; size_t n = 0; double cyc = 0, wall = 0;
        for (size_t i = 0; i < nblk; ++i) {
            const unsigned long long* t = &ht[i * 16];
            if (!t[5]) continue;
            for (int j = 0; j < 9; ++j) d[j] += (double)(t[order[j + 1]] - t[order[j]]);
            cyc += (double)(t[5] - t[0]); wall += (double)(t[12] - t[13]);
            ++n;
        }
        const double clk = cyc / (wall * 0.01);   // shader cycles per us, measured (wall_clock64 = 100 MHz)
        printf("   shader clock %.2f GHz;", clk / 1e3);
        double life = 0;
        printf(" per block, us:");
        for (int j = 0; j < 9; ++j) { printf(" %s %.2f |", names[j], d[j] / n / clk); life += d[j] / n / clk; }
        printf(" life %.2f; %.2f blocks in flight per CU\n", life, life * n / us / 256.0);
        hipFree(tim);
#endif
        hipFree(x); hipFree(y); hipFree(w1); hipFree(w2); hipFree(b);
    }
    return 0;
}
