// Developer probe (GPU): does it matter WHICH register file the streamed A operand lands in?  fp32 MFMA 32x32x2,
// 4 waves (1/SIMD), 128 rows x 32 frames per wave, 6-stage A prefetch (4 x dwordx4 per 16 MFMA), B = ds_read_b128.
//   MODE 0  A -> ArchVGPR (asm loads),  acc in AccVGPR      (what hipcc generates for the builtin path)
//   MODE 1  A -> AccVGPR  (asm loads),  acc in AccVGPR
//   MODE 2  A -> AccVGPR,               acc in ArchVGPR
//   MODE 3  A -> ArchVGPR,              acc in ArchVGPR
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe4.bin tools/mfma_probe4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ void load4(f32x4& dst, const float4* p) {
    if (MODE == 1 || MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(dst) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int MODE>
__device__ __forceinline__ void mma(f32x16& acc, float a, float b) {
    if (MODE == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    if (MODE == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
    if (MODE == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
    if (MODE == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int MODE, int NLOAD>      // NLOAD: A loads actually issued per chunk (4 = real, 0 = none: MFMA + B only)
__global__ __launch_bounds__(256, 1) void probe(const float4* __restrict__ w, float* out, unsigned long long* cyc, int rounds) {
    constexpr int S = 6, LDF = 260;
    __shared__ __attribute__((aligned(16))) float lds[48 * LDF];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 48 * LDF; i += 256) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float4* ap = w + (size_t)wv * (96 * 256) + lane;
    f32x4 a[S][4];
    float4 b[2];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) load4<MODE>(a[i][m], ap + i * 256 + m * 64);
    b[0] = *reinterpret_cast<const float4*>(lds + (j + 8) * LDF + 4 * h);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SB();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int kc = it * S + u;
            const float4* p = ap + (size_t)((kc + S - 1) & 63) * 256;
            const int kn = kc + 1;
            b[(u + 1) & 1] = *reinterpret_cast<const float4*>(lds + (j + 8 + ((kn >> 5) & 1) - 1) * LDF + (kn & 31) * 8 + 4 * h);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float4 bv4 = b[u & 1];
                    const float bv = s == 0 ? bv4.x : s == 1 ? bv4.y : s == 2 ? bv4.z : bv4.w;
                    mma<MODE>(acc[m], a[u][m][s], bv);
                    // one A load for chunk kc+S-1 behind each of the first four MFMAs (it overwrites stage (u+S-1)%S,
                    // whose last reader was chunk kc-1)
                    if (s == 0 && m < NLOAD) load4<MODE>(a[(u + S - 1) % S][m], p + m * 64);
                }
            // chunk kc+1's loads were issued S-2 chunks ago: allow (S-2)*4 + 4 newer ones in flight
            if (NLOAD) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * 4) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) sum += acc[m][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
}

template <int MODE, int NLOAD>
int run(const char* name, const float4* w, float* out, unsigned long long* cyc, int blocks) {
    const int rounds = 160;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<MODE, NLOAD><<<blocks, 256>>>(w, out, cyc, rounds);
    CK(hipEventRecord(e0));
    probe<MODE, NLOAD><<<blocks, 256>>>(w, out, cyc, rounds);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0; unsigned long long mx = 0;
    for (auto v : h) { mean += (double)v; mx = v > mx ? v : mx; }
    mean /= h.size();
    const double mf = (double)rounds * 6 * 16;
    printf("%-52s cycles/MFMA mean %6.2f max %6.2f  kernel %.3f ms -> %.1f TFLOP/s, %.2f GHz\n", name, mean / mf, mx / mf, ms,
           blocks * 4 * mf * 4096.0 / (ms * 1e-3) / 1e12, mean / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    float4* w; float* out; unsigned long long* cyc;
    CK(hipMalloc(&w, (size_t)4 * 96 * 256 * 16 * 2));
    CK(hipMemset(w, 0, (size_t)4 * 96 * 256 * 16 * 2));
    CK(hipMalloc(&out, 4096 * 512 * 4));
    CK(hipMalloc(&cyc, 4096 * 8 * 8));
    const int blocks = 256;
    run<0, 0>("A none            acc Acc   (MFMA + B only)", w, out, cyc, blocks);
    run<0, 4>("A -> ArchVGPR     acc Acc", w, out, cyc, blocks);
    run<1, 4>("A -> AccVGPR      acc Acc", w, out, cyc, blocks);
    run<2, 4>("A -> AccVGPR      acc Arch", w, out, cyc, blocks);
    run<3, 4>("A -> ArchVGPR     acc Arch", w, out, cyc, blocks);
    run<0, 2>("A -> ArchVGPR x2  acc Acc   (half the A loads)", w, out, cyc, blocks);
    return 0;
}
