// mfma_split_probe.hip - developer probe (GPU, stand-alone: hipcc, no torch): is an fp32-accurate GEMM on the BF16 matrix pipe worth building?
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  Splitting every fp32 operand into three bf16 planes
// (x = x0 + x1 + x2, 8 + 8 + 8 mantissa bits: exact for normal fp32) and accumulating the six products with i + j <= 2 in fp32 on
// v_mfma_f32_32x32x16_bf16 keeps ~2^-24 relative accuracy per product at 6 x 32 = 192 cycles per K = 16 instead of 8 x 64 = 512.
// This probe measures, on one 32 x 32 tile with K = 768 (the dilated conv of the denoiser):
//   accuracy    max / rms error against an fp64 host reference of  (a) the fp32 MFMA chain  (b) the 6-product split  (c) the 3-product split
//   throughput  cycles per K = 16 block of (a) and (b) from registers only (pure matrix-pipe rate), and of (b) with the B operand split
//               in the loop (v_cvt + v_sub fillers beside the MFMAs)
// build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_split_probe tools/mfma_split_probe.hip && /tmp/mfma_split_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 768;

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }                 // round to nearest even
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
    a = to_bf16(x);
    const float r1 = x - (float)a;
    b = to_bf16(r1);
    c = to_bf16(r1 - (float)b);
}

// A [32][K], B [K][32] row-major fp32; D [32][32].  One wave.  mode 0: fp32 MFMA; 1: six-product split; 2: three-product split
__global__ void k_accuracy(const float* A, const float* B, float* D, int mode) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a0, a1, a2, b0, b1, b2;
            for (int e = 0; e < 8; ++e) {                                                   // lane (i, h): A[i][k + 8 h + e], B[k + 8 h + e][i]
                __bf16 x, y, z;
                split3(A[i * K + k + 8 * h + e], x, y, z); a0[e] = x; a1[e] = y; a2[e] = z;
                split3(B[(k + 8 * h + e) * 32 + i], x, y, z); b0[e] = x; b1[e] = y; b2[e] = z;
            }
            // smallest terms first
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

// throughput: `iters` K = 16 blocks per wave, 4 independent accumulators, operands from registers
__global__ __launch_bounds__(256) void k_rate(float* out, uint64_t* cycles, int iters, int mode) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float af = 1.0f + lane * 1e-3f, bf = 0.5f - lane * 1e-3f;
    bf16x8 a0, a1, a2, b0, b1, b2;
    for (int e = 0; e < 8; ++e) { a0[e] = to_bf16(af + e); a1[e] = to_bf16(af * 1e-3f); a2[e] = to_bf16(af * 1e-6f); b0[e] = to_bf16(bf); b1[e] = to_bf16(bf * 1e-3f); b2[e] = to_bf16(bf * 1e-6f); }
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[q], 0, 0, 0);
        } else {
            if (mode == 2) {                                                                // split the B fragment here: 8 values -> 3 planes
#pragma unroll
                for (int e = 0; e < 8; ++e) { __bf16 x, y, z; split3(bf + (float)(it & 7) + e, x, y, z); b0[e] = x; b1[e] = y; b2[e] = z; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[q], 0, 0, 0);
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// weight-stream ceiling: every workgroup (4 waves, one per SIMD) walks the SAME `bytes`-long buffer front to back, 1 KiB per wave per load
// (lane * 16 B), `unroll` loads in flight - the access pattern of the layer kernels' A operand (all CUs of an XCD in near lock-step on
// one weight stream).  Aggregate GB/s = what the L2s can hand the CUs for such a stream; the split scheme needs ~4x today's rate.
__global__ __launch_bounds__(256) void k_wstream(const float4* w, float* out, size_t n4_per_wave_pass, int passes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ps = 0; ps < passes; ++ps) {
        const float4* base = w + (size_t)wave * n4_per_wave_pass + lane;                  // each wave its own quarter (its own rows)
#pragma unroll 8
        for (size_t i = 0; i < n4_per_wave_pass; i += 64) {
            const float4 v = base[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    std::vector<float> hA(32 * K), hB(K * 32);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (auto& v : hA) v = rnd() * 0.06f;                                                   // ~ fan-in scaled weights
    for (auto& v : hB) v = rnd() * 1.7f;                                                    // ~ activations
    std::vector<double> ref(32 * 32, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[i * K + k] * (double)hB[k * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dD;
    CK(hipMalloc((void**)&dA, hA.size() * 4)); CK(hipMalloc((void**)&dB, hB.size() * 4)); CK(hipMalloc((void**)&dD, 32 * 32 * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const char* names[3] = {"fp32 MFMA 32x32x2", "bf16 x6 split (32x32x16)", "bf16 x3 split (32x32x16)"};
    double scale = 0; for (double v : ref) scale = std::fmax(scale, std::fabs(v));
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k_accuracy, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode);
        std::vector<float> hD(32 * 32);
        CK(hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost));
        double mx = 0, ss = 0;
        for (int q = 0; q < 32 * 32; ++q) { const double e = std::fabs((double)hD[q] - ref[q]); mx = std::fmax(mx, e); ss += e * e; }
        std::printf("{\"probe\": \"accuracy\", \"impl\": \"%s\", \"K\": %d, \"max_abs_err\": %.3e, \"rms_err\": %.3e, \"max_abs_ref\": %.3e, \"max_err_rel_to_max\": %.3e}\n",
                    names[mode], K, mx, std::sqrt(ss / 1024), scale, mx / scale);
    }
    float* dout; uint64_t* dcyc;
    const int blocks = 256, iters = 2000;
    CK(hipMalloc((void**)&dout, blocks * 256 * 4)); CK(hipMalloc((void**)&dcyc, blocks * 8));
    const char* rnames[3] = {"fp32 MFMA: 8 x 32x32x2 per K=16", "bf16 x6: 6 x 32x32x16 per K=16 (operands in registers)", "bf16 x6 + the B fragment split in the loop"};
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, dout, dcyc, iters, mode);     // warm
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, dout, dcyc, iters, mode);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> hc(blocks);
        CK(hipMemcpy(hc.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost));
        double cyc = 0; for (auto c : hc) cyc += (double)c; cyc /= blocks;
        const double flop_equiv = (double)blocks * 4 /*waves*/ * iters * 4 /*acc*/ * 2.0 * 32 * 32 * 16;   // fp32-equivalent FLOPs
        std::printf("{\"probe\": \"rate\", \"impl\": \"%s\", \"ms\": %.3f, \"fp32_equivalent_tflops\": %.1f, \"counter_ticks_per_K16_block_per_accumulator\": %.1f}\n",
                    rnames[mode], ms, flop_equiv / (ms * 1e-3) / 1e12, cyc / iters / 4);
    }
    // weight-stream ceiling: 2 MB (one layer's fp32 weights) and 3 MB (the same as three bf16 planes), 256 workgroups = one per CU
    for (size_t mb : {2, 3}) {
        const size_t bytes = mb << 20, n4 = bytes / 16, per_wave = n4 / 4;
        float4* dw; CK(hipMalloc((void**)&dw, bytes)); CK(hipMemset(dw, 0, bytes));
        const int passes = 200;
        hipLaunchKernelGGL(k_wstream, dim3(256), dim3(256), 0, 0, dw, dout, per_wave, 2);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_wstream, dim3(256), dim3(256), 0, 0, dw, dout, per_wave, passes);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double total = (double)bytes * passes * 256;
        std::printf("{\"probe\": \"weight_stream\", \"buffer_MB\": %zu, \"workgroups\": 256, \"ms\": %.3f, \"aggregate_TBps\": %.2f, \"GBps_per_CU\": %.1f, "
                    "\"note\": \"k_loop streams 2 MB per layer per CU in ~63 us (32 GB/s/CU, 8.2 TB/s aggregate); the 6-product split needs 3 MB in ~25 us (120 GB/s/CU)\"}\n",
                    mb, ms, total / (ms * 1e-3) / 1e12, (double)bytes * passes / (ms * 1e-3) / 1e9);
        CK(hipFree(dw));
    }
    return 0;
}
