// Developer probe (GPU), round 6: could the vocoder's folded stages (8 / 16 channels on a 32-row MFMA block: the F-fold carries K + F - 1 taps for K)
// run on a 16-ROW matrix shape instead?  v_mfma_f32_16x16x1_4B_f32 = four independent 16 x 16 x 1 blocks per instruction: 64 columns x 16 rows,
// one k.  Two questions: (1) does it issue at the same FLOP rate as the 32x32x2 form (2048 FLOP in 32 cycles)?  (2) does its accumulation round
// like the 32x32x2 form's, i.e. would a kernel built on it stay BIT-IDENTICAL to the one-convolution kernels (c + a0 b0 + a1 b1 in k order)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_probe tools/mfma_shape_probe.hip && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void rate(float* out, unsigned long long* cyc, int rounds) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float a = 0.5f + lane, b = 0.001f * lane;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (SHAPE == 0) asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
}

// one wave: D32 = C + A[32 x 2] B[2 x 32] on the 32x32x2 form; D16 = the same for rows / columns < 16 as two 16x16x1 steps (k = 0, then k = 1)
__global__ void ident(const float* A, const float* B, const float* C, float* d32, float* d16) {
    const int l = threadIdx.x;
    f32x16 c32;
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31; c32[r] = C[row * 32 + col]; }
    c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c32, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31; d32[row * 32 + col] = c32[r]; }
    // block 0 of the 4-block form: A lane l: row l % 16 (block l / 16); only block 0's values matter here
    f32x16 c16;
    for (int r = 0; r < 16; ++r) { const int row = 4 * (l >> 4) + (r & 3), col = l & 15; c16[r] = (r < 4) ? C[row * 32 + col] : 0.f; }
    for (int k = 0; k < 2; ++k)
        c16 = __builtin_amdgcn_mfma_f32_16x16x1f32(A[(l & 15) * 2 + k], B[k * 32 + (l & 15)], c16, 0, 0, 0);
    if (l < 64) for (int r = 0; r < 4; ++r) { const int row = 4 * (l >> 4) + r, col = l & 15; if ((l >> 4) < 4) d16[row * 32 + col] = c16[r]; }
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 4 * 8));
    for (int shape = 0; shape < 2; ++shape) {
        const int rounds = 2000;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        if (shape == 0) rate<0><<<256, 256>>>(out, cyc, rounds); else rate<1><<<256, 256>>>(out, cyc, rounds);
        CK(hipEventRecord(e0));
        if (shape == 0) rate<0><<<256, 256>>>(out, cyc, rounds); else rate<1><<<256, 256>>>(out, cyc, rounds);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(1024); CK(hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (auto v : h) mean += (double)v; mean /= 1024;
        const double mf = rounds * 16.0, flop = shape == 0 ? 2048.0 : 4096.0;
        printf("{\"mfma\": \"%s\", \"cycles_per_mfma\": %.2f, \"flop_per_cycle_per_simd\": %.1f, \"tflops\": %.1f}\n", shape == 0 ? "16x16x1_4B_f32" : "32x32x2_f32",
               mean / mf, flop / (mean / mf), 1024 * mf * flop / (ms * 1e-3) / 1e12);
    }
    float *A, *B, *C, *d32, *d16;
    CK(hipMalloc(&A, 64 * 4)); CK(hipMalloc(&B, 64 * 4)); CK(hipMalloc(&C, 1024 * 4)); CK(hipMalloc(&d32, 1024 * 4)); CK(hipMalloc(&d16, 1024 * 4));
    long mism = 0, total = 0;
    srand(7);
    for (int trial = 0; trial < 2000; ++trial) {
        std::vector<float> hA(64), hB(64), hC(1024), h32(1024), h16(1024, 0.f);
        const float scale = (trial % 4 == 0) ? 1e-3f : (trial % 4 == 1) ? 1.f : (trial % 4 == 2) ? 37.f : 1e4f;
        for (auto& v : hA) v = scale * ((float)rand() / RAND_MAX - 0.5f);
        for (auto& v : hB) v = (float)rand() / RAND_MAX - 0.5f;
        for (auto& v : hC) v = (trial % 3 == 0) ? 0.f : scale * 3.f * ((float)rand() / RAND_MAX - 0.5f);
        CK(hipMemcpy(A, hA.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(C, hC.data(), 4096, hipMemcpyHostToDevice));
        CK(hipMemset(d16, 0, 4096));
        ident<<<1, 64>>>(A, B, C, d32, d16);
        CK(hipMemcpy(h32.data(), d32, 4096, hipMemcpyDeviceToHost)); CK(hipMemcpy(h16.data(), d16, 4096, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { ++total; if (memcmp(&h32[i * 32 + j], &h16[i * 32 + j], 4)) ++mism; }
    }
    printf("{\"bit_identity_32x32x2_vs_two_16x16x1_steps\": {\"elements\": %ld, \"mismatches\": %ld}}\n", total, mism);
    return 0;
}
