// Developer probe (GPU): does operand-load traffic slow fp32 MFMA issue, and does the 16x16x4 shape (half the
// accumulator write-back per FLOP of 32x32x2) leave room for it?
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe2.bin tools/mfma_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// SHAPE 0: 32x32x2 (4 accumulators of 16 regs), chunk = 8 k: 16 MFMA, 4 x dwordx4 A loads, 4 B dwords
// SHAPE 1: 16x16x4 (16 accumulators of 4 regs), chunk = 16 k: 64 MFMA, 8 x dwordx4 A loads, 8 B dwords
// LOADS: bit0 global A loads, bit1 LDS B loads
template <int SHAPE, int LOADS>
__global__ __launch_bounds__(256, 1) void probe(const float4* __restrict__ w, float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[256 * 48];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 256 * 48; i += 256) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    constexpr int NA = SHAPE ? 8 : 4;         // float4 A loads per chunk
    constexpr int NBR = SHAPE ? 8 : 4;        // B dwords per chunk
    constexpr int KCH = SHAPE ? 48 : 96;      // chunks per 768-deep K
    f32x16 acc32[4];
    f32x4 acc16[16];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc32[m][r] = 0.f;
    for (int m = 0; m < 16; ++m) for (int r = 0; r < 4; ++r) acc16[m][r] = 0.f;
    const float4* ap = w + wv * (96 * 256) + lane;
    const float* bp = SHAPE ? lds + (lane >> 4) * 48 + 8 + (lane & 15) : lds + (lane >> 5) * 4 * 48 + 8 + (lane & 31);
    float4 a[2][NA];
    float b[2][NBR];
    for (int m = 0; m < NA; ++m) a[0][m] = ap[m * 64];
    for (int s = 0; s < NBR; ++s) b[0][s] = bp[s * 48];
    SB();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kc = (it * 2 + u + 1) % KCH;
            if (LOADS & 1) { for (int m = 0; m < NA; ++m) a[u ^ 1][m] = ap[(size_t)kc * (NA * 64) + m * 64]; }
            else { for (int m = 0; m < NA; ++m) a[u ^ 1][m] = a[u][m]; }
            if (LOADS & 2) { for (int s = 0; s < NBR; ++s) b[u ^ 1][s] = bp[(kc & 15) * 16 * 48 + (s & 3) * 4 * 48 + (s >> 2) * 16]; }
            else { for (int s = 0; s < NBR; ++s) b[u ^ 1][s] = b[u][s]; }
            if (SHAPE == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float av = s == 0 ? a[u][m].x : s == 1 ? a[u][m].y : s == 2 ? a[u][m].z : a[u][m].w;
                        acc32[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][s], acc32[m], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const float av = s == 0 ? a[u][m].x : s == 1 ? a[u][m].y : s == 2 ? a[u][m].z : a[u][m].w;
#pragma unroll
                        for (int f = 0; f < 2; ++f)
                            acc16[m * 2 + f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u][f * 4 + s], acc16[m * 2 + f], 0, 0, 0);
                    }
            }
            // one load per MFMA gap
            if (LOADS & 1)
                for (int i = 0; i < NA; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            if (LOADS & 2)
                for (int i = 0; i < NBR / 2; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            SB();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) sum += acc32[m][r];
    for (int m = 0; m < 16; ++m) for (int r = 0; r < 4; ++r) sum += acc16[m][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
}

template <int SHAPE, int LOADS>
int run(const char* name, const float4* w, float* out, unsigned long long* cyc, int blocks) {
    const int iters = SHAPE ? 240 : 480;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<SHAPE, LOADS><<<blocks, 256>>>(w, out, cyc, iters);
    CK(hipEventRecord(e0));
    probe<SHAPE, LOADS><<<blocks, 256>>>(w, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0; unsigned long long mx = 0;
    for (auto v : h) { mean += (double)v; mx = v > mx ? v : mx; }
    mean /= h.size();
    const double mf = (double)iters * 2 * (SHAPE ? 64 : 16);
    const double flop = SHAPE ? 2048.0 : 4096.0;
    printf("%-34s blocks=%4d  cycles/MFMA mean %6.2f max %6.2f  (x%.3f of issue rate)  kernel %.3f ms -> %.1f TFLOP/s, %.2f GHz\n", name, blocks,
           mean / mf, mx / mf, mean / mf / (SHAPE ? 32.0 : 64.0), ms, blocks * 4 * mf * flop / (ms * 1e-3) / 1e12, mean / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    float4* w; float* out; unsigned long long* cyc;
    CK(hipMalloc(&w, (size_t)4 * 96 * 256 * 16 * 2));
    CK(hipMemset(w, 0, (size_t)4 * 96 * 256 * 16 * 2));
    CK(hipMalloc(&out, 4096 * 256 * 4));
    CK(hipMalloc(&cyc, 4096 * 4 * 8));
    for (int blocks : {256, 512}) {
        run<0, 0>("32x32x2 only", w, out, cyc, blocks);
        run<0, 1>("32x32x2 + global A", w, out, cyc, blocks);
        run<0, 2>("32x32x2 + LDS B", w, out, cyc, blocks);
        run<0, 3>("32x32x2 + global A + LDS B", w, out, cyc, blocks);
        run<1, 0>("16x16x4 only", w, out, cyc, blocks);
        run<1, 1>("16x16x4 + global A", w, out, cyc, blocks);
        run<1, 2>("16x16x4 + LDS B", w, out, cyc, blocks);
        run<1, 3>("16x16x4 + global A + LDS B", w, out, cyc, blocks);
    }
    return 0;
}
