// dsd_example.cpp - the C ABI of libdsdenoise.so (include/dsd.h) from a host that is neither Python nor torch: the complete
// K-step DDPM sampling of the DiffSpeech denoiser on synthetic weights, conditioner and noise, HIP runtime API only.
//
//   hipcc -O2 -I include examples/dsd_example.cpp -L diffsinger_amd -ldsdenoise -Wl,-rpath,$PWD/diffsinger_amd -o dsd_example
//   ./dsd_example [B] [T] [K]          (default 8 1024 100)  ->  one JSON line: mel checksum, wall time, mel-frames/s
//
// What a maintainer of a C / C++ / Go / Rust host would write against the same header.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dsd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define DSD_OK_(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #x, r_, dsd_last_error()); return 3; } } while (0)

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static float urand() {                                   // xorshift64*, uniform in (-1, 1)
    g_state ^= g_state >> 12; g_state ^= g_state << 25; g_state ^= g_state >> 27;
    return (float)((double)((g_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
static float nrand() {                                   // Box-Muller
    const float u = 0.5f * (urand() + 1.f) + 1e-7f, v = 0.5f * (urand() + 1.f);
    return std::sqrt(-2.f * std::log(u)) * std::cos(6.2831853f * v);
}

static std::vector<void*> g_allocs;
static const float* upload(const std::vector<float>& h) {
    void* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    g_allocs.push_back(d);
    return static_cast<const float*>(d);
}
static const float* weight(size_t n, size_t fan_in, float gain = 1.f) {
    std::vector<float> h(n);
    const float s = gain * std::sqrt(3.f / (float)fan_in);
    for (auto& v : h) v = s * urand();
    return upload(h);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? std::atoi(argv[1]) : 8, T = argc > 2 ? std::atoi(argv[2]) : 1024, K = argc > 3 ? std::atoi(argv[3]) : 100;
    const int M = 80, C = 256, H = 256, L = 20;
    if (dsd_abi_version() != DSD_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    HIP_OK(hipSetDevice(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    dsd_config cfg{M, C, H, L, 1};                        // DiffSpeech: residual_channels 256, 20 layers, dilation_cycle_length 1
    dsd_handle* h = nullptr;
    DSD_OK_(dsd_create(&cfg, 0, &h));

    // parameters in torch state_dict layout (usr/diff/net.py:91-105), synthetic
    dsd_weights w{};
    w.input_projection_w = weight((size_t)C * M, M); w.input_projection_b = weight(C, C);
    w.mlp0_w = weight((size_t)4 * C * C, C); w.mlp0_b = weight(4 * C, 4 * C);
    w.mlp2_w = weight((size_t)C * 4 * C, 4 * C); w.mlp2_b = weight(C, C);
    std::vector<const float*> dw(L), db(L), pw(L), pb(L), cw(L), cb(L), ow(L), ob(L);
    for (int l = 0; l < L; ++l) {
        dw[l] = weight((size_t)2 * C * C * 3, 3 * C); db[l] = weight(2 * C, 2 * C);
        pw[l] = weight((size_t)C * C, C); pb[l] = weight(C, C);
        cw[l] = weight((size_t)2 * C * H, H); cb[l] = weight(2 * C, 2 * C);
        ow[l] = weight((size_t)2 * C * C, C); ob[l] = weight(2 * C, 2 * C);
    }
    w.dilated_conv_w = dw.data(); w.dilated_conv_b = db.data();
    w.diffusion_projection_w = pw.data(); w.diffusion_projection_b = pb.data();
    w.conditioner_projection_w = cw.data(); w.conditioner_projection_b = cb.data();
    w.output_projection_w = ow.data(); w.output_projection_b = ob.data();
    w.skip_projection_w = weight((size_t)C * C, C); w.skip_projection_b = weight(C, C);
    w.final_projection_w = weight((size_t)M * C, C, 0.1f); w.final_projection_b = weight(M, M);
    DSD_OK_(dsd_load_weights(h, &w, stream));

    std::vector<double> betas(K);                         // linear_beta_schedule(K, max_beta = 0.06), shallow_diffusion_tts.py:44-49
    for (int i = 0; i < K; ++i) betas[i] = 1e-4 + (0.06 - 1e-4) * i / (K > 1 ? K - 1 : 1);
    DSD_OK_(dsd_set_schedule(h, betas.data(), K));
    std::vector<float> smin(M, -6.f), smax(M, 0.5f);
    DSD_OK_(dsd_set_spec_range(h, smin.data(), smax.data()));

    std::vector<float> hc((size_t)B * H * T), hx((size_t)B * M * T);
    for (auto& v : hc) v = nrand();
    for (auto& v : hx) v = nrand();
    const float* cond = upload(hc);                       // [B][H][T] contiguous
    float* x = const_cast<float*>(upload(hx));            // x_T, [B][M][T]
    float* mel = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&mel), (size_t)B * T * M * sizeof(float)));
    if (!cond || !x) { std::fprintf(stderr, "hipMalloc failed\n"); return 2; }

    DSD_OK_(dsd_prepare(h, B, T, cond, (int64_t)H * T, T, 1, stream));
    DSD_OK_(dsd_set_noise_seed(h, 1234));                 // noise == NULL below: Philox draws inside the sampler epilogue
    HIP_OK(hipStreamSynchronize(stream));
    const auto t0 = std::chrono::steady_clock::now();
    DSD_OK_(dsd_sample_ddpm(h, x, nullptr, K, stream));   // the whole reverse loop: one persistent kernel
    DSD_OK_(dsd_denorm_spec(h, x, nullptr, mel, stream)); // -> [B][T][M]
    HIP_OK(hipStreamSynchronize(stream));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<float> hm((size_t)B * T * M);
    HIP_OK(hipMemcpy(hm.data(), mel, hm.size() * sizeof(float), hipMemcpyDeviceToHost));
    double sum = 0, sumabs = 0;
    bool finite = true;
    for (float v : hm) { sum += v; sumabs += std::fabs(v); finite = finite && std::isfinite(v); }
    std::printf("{\"B\": %d, \"T\": %d, \"K\": %d, \"seconds\": %.6f, \"mel_frames_per_s\": %.1f, \"mel_sum\": %.6e, \"mel_mean_abs\": %.6e, "
                "\"finite\": %s, \"loop_mode\": %d, \"device_bytes\": %lld}\n",
                B, T, K, sec, (double)B * T / sec, sum, sumabs / hm.size(), finite ? "true" : "false", dsd_get_loop_mode(h),
                (long long)dsd_device_bytes(h));
    dsd_destroy(h);
    for (void* p : g_allocs) (void)hipFree(p);
    (void)hipFree(mel);
    (void)hipStreamDestroy(stream);
    return finite ? 0 : 4;
}
