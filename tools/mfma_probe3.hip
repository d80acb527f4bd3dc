// Developer probe (GPU): inner-loop geometry candidates for the residual-layer kernel, one 32-frame tile per CU.
//   WAVES = 4: 1 wave/SIMD, 128 rows x 32 frames per wave (4 A loads + B per 16 MFMA)
//   WAVES = 8: 2 waves/SIMD, 64 rows x 32 frames per wave (2 A loads + B per 8 MFMA) - the partner wave's MFMAs
//              fill the issue bubbles of this wave's loads
//   BMODE 0: B as 4 ds_read_b32 from [channel][frame];  1: one ds_read_b128 from [frame][channel+pad]
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe3.bin tools/mfma_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int WAVES, int BMODE, int STAGES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe(const float4* __restrict__ w, float* out, unsigned long long* cyc, int rounds) {
    constexpr int NMB = 16 / WAVES;
    constexpr int LDF = 260;                       // [frame][channel] row stride
    constexpr int LDC = 48;                        // [channel][frame] row stride
    __shared__ __attribute__((aligned(16))) float lds[48 * LDF];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 48 * LDF; i += WAVES * 64) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc[NMB];
    for (int m = 0; m < NMB; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    // packed A: [wave][kc 96][mb NMB][lane 64] float4 ; all blocks stream the same 1.5 MB (one layer's conv weights)
    const float4* ap = w + (size_t)wv * (96 * NMB * 64) + lane;
    float4 a[STAGES][NMB];
    float4 b[2];
    auto lda = [&](float4 (&dst)[NMB], int kc) {
        const float4* p = ap + (size_t)(kc & 63) * (NMB * 64);
#pragma unroll
        for (int m = 0; m < NMB; ++m) dst[m] = p[m * 64];
    };
    auto ldb = [&](float4& dst, int kc) {
        const int k8 = (kc & 31) * 8, tap = (kc >> 5) & 1;
        if (BMODE == 1) {
            dst = *reinterpret_cast<const float4*>(lds + (j + 8 + (tap - 1)) * LDF + k8 + 4 * h);
        } else {
            const float* p = lds + (k8 % 248 + 4 * h) * LDC + 8 + (tap - 1) + j;
            dst = make_float4(p[0], p[LDC], p[2 * LDC], p[3 * LDC]);
        }
    };
#pragma unroll
    for (int i = 0; i < STAGES - 1; ++i) lda(a[i], i);
    ldb(b[0], 0);
    SB();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int u = 0; u < STAGES * 2; ++u) {     // STAGES*2 is a multiple of both rotation periods
            const int kc = it * STAGES * 2 + u;
            lda(a[(u + STAGES - 1) % STAGES], kc + STAGES - 1);
            ldb(b[(u + 1) & 1], kc + 1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < NMB; ++m) {
                    const float4 av4 = a[u % STAGES][m];
                    const float4 bv4 = b[u & 1];
                    const float av = s == 0 ? av4.x : s == 1 ? av4.y : s == 2 ? av4.z : av4.w;
                    const float bv = s == 0 ? bv4.x : s == 1 ? bv4.y : s == 2 ? bv4.z : bv4.w;
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
                }
            for (int i = 0; i < NMB; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            for (int i = 0; i < (BMODE ? 1 : 2); ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            SB();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int m = 0; m < NMB; ++m) for (int r = 0; r < 16; ++r) sum += acc[m][r];
    out[blockIdx.x * (WAVES * 64) + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * WAVES + wv] = t1 - t0;
}

template <int WAVES, int BMODE, int STAGES>
int run(const char* name, const float4* w, float* out, unsigned long long* cyc, int blocks) {
    const int rounds = 960 / (STAGES * 2);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<WAVES, BMODE, STAGES><<<blocks, WAVES * 64>>>(w, out, cyc, rounds);
    CK(hipEventRecord(e0));
    probe<WAVES, BMODE, STAGES><<<blocks, WAVES * 64>>>(w, out, cyc, rounds);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * WAVES);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0; unsigned long long mx = 0;
    for (auto v : h) { mean += (double)v; mx = v > mx ? v : mx; }
    mean /= h.size();
    const double mf_wave = (double)rounds * STAGES * 2 * 4 * (16 / WAVES);     // MFMAs per wave
    const double mf_simd = mf_wave * (WAVES / 4);                               // MFMAs per SIMD
    printf("%-44s blocks=%4d  SIMD cycles/MFMA mean %6.2f max %6.2f  kernel %.3f ms -> %.1f TFLOP/s, %.2f GHz\n", name, blocks,
           mean / mf_simd, mx / mf_simd, ms, blocks * 4 * mf_simd * 4096.0 / (ms * 1e-3) / 1e12, mean / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    float4* w; float* out; unsigned long long* cyc;
    CK(hipMalloc(&w, (size_t)4 * 96 * 256 * 16 * 2));
    CK(hipMemset(w, 0, (size_t)4 * 96 * 256 * 16 * 2));
    CK(hipMalloc(&out, 4096 * 512 * 4));
    CK(hipMalloc(&cyc, 4096 * 8 * 8));
    const int blocks = 256;
    run<4, 0, 6>("4 waves, B 4x b32, 6 stages", w, out, cyc, blocks);
    run<4, 1, 6>("4 waves, B b128, 6 stages", w, out, cyc, blocks);
    run<8, 0, 6>("8 waves, B 4x b32, 6 stages", w, out, cyc, blocks);
    run<8, 1, 6>("8 waves, B b128, 6 stages", w, out, cyc, blocks);
    run<8, 1, 3>("8 waves, B b128, 3 stages", w, out, cyc, blocks);
    run<8, 1, 12>("8 waves, B b128, 12 stages", w, out, cyc, blocks);
    return 0;
}
