// Do the three ResBlock branches (k = 3, 7, 11) of one HiFi-GAN stage run faster on three streams than back to
// back on one?  (tools only)   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_concurrent.hip
#include "../tts-arabic-pytorch_amd/csrc/conv_mfma.hip"
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
namespace ttsamd {
static thread_local std::string g_err;
int32_t launch_conv_bf16_any(const ConvParams&, hipStream_t) { return -1; }
void set_error(const char* fmt, ...) { char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); g_err = buf; fprintf(stderr, "ERR %s\n", buf); }
}
using namespace ttsamd;
struct Conv { ConvParams p; };
static Conv make(int B, int C, int k, int dil, int L, const float* x, float* y) {
    Conv c; std::memset(&c.p, 0, sizeof c.p);
    const int cp = cout_padded(C);
    float *w, *b; size_t nw = (size_t)C * k * cp;
    hipMalloc(&w, nw * 4); hipMalloc(&b, cp * 4);
    std::vector<float> hw(nw);
    for (auto& v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.05f;
    hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice); hipMemset(b, 0, cp * 4);
    ConvParams& p = c.p;
    p.x = x; p.x_bs = (int64_t)C * L; p.x_cs = L; p.w = w; p.bias = b;
    p.y = y; p.y_bs = (int64_t)C * L; p.y_cs = L; p.y_ts = 1; p.res = x; p.r_bs = p.x_bs; p.r_cs = L;
    p.len_in_mul = p.len_out_mul = 1; p.Lin = p.Nout = L; p.Cin = C; p.Cout = C; p.CoutP = cp; p.K = k;
    p.dil = dil; p.pad = (k * dil - dil) / 2; p.n_phase = 1; p.in_slope = 0.1f; p.div = 1.f; p.batch = B;
    return c;
}
int main() {
    struct St { int C, L; } stages[] = {{256, 3584}, {128, 28672}, {64, 57344}, {32, 114688}};
    hipStream_t st[3];
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (auto sg : stages) {
        const int B = 32;
        size_t n = (size_t)B * sg.C * sg.L;
        float *x, *y[3];
        hipMalloc(&x, n * 4);
        std::vector<float> hx(1 << 20);
        for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
        for (size_t o = 0; o < n; o += hx.size()) hipMemcpy(x + o, hx.data(), std::min(hx.size(), n - o) * 4, hipMemcpyHostToDevice);
        for (auto& p : y) hipMalloc(&p, n * 4);
        const int ks[3] = {3, 7, 11};
        Conv c[3][2];
        for (int i = 0; i < 3; ++i) { c[i][0] = make(B, sg.C, ks[i], 3, sg.L, x, y[i]); c[i][1] = make(B, sg.C, ks[i], 1, sg.L, x, y[i]); }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto seq = [&] { for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) launch_conv(c[i][j].p, st[0]); };
        auto par = [&] { for (int j = 0; j < 2; ++j) for (int i = 0; i < 3; ++i) launch_conv(c[i][j].p, st[i]); };
        float ms_seq, ms_par;
        seq(); hipDeviceSynchronize();
        hipEventRecord(e0, st[0]); for (int r = 0; r < 5; ++r) seq(); hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_seq, e0, e1);
        par(); hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 5; ++r) par();
        hipDeviceSynchronize();
        ms_par = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        double fl = 2.0 * sg.C * sg.C * (3 + 7 + 11) * 2 * (double)B * sg.L * 5;
        printf("C=%3d L=%6d: one stream %.3f ms (%.1f TF)   three streams %.3f ms (%.1f TF)\n", sg.C, sg.L, ms_seq / 5, fl / ms_seq / 1e9,
               ms_par / 5, fl / ms_par / 1e9);
        hipFree(x); for (auto& p : y) hipFree(p);
    }
    return 0;
}
