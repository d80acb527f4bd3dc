// Developer probe (GPU), round 6: WHO should fetch the A stream of an fp32 MFMA loop, and into what?
// tools/mfma_filler_probe.hip priced a 1 KiB `buffer_load_dwordx4` issued by the computing wave itself at ~10-16 cycles of matrix time (8 % of
// k_loop_wino's contraction: 4 loads per 16 short MFMAs) while a `ds_read_b128` costs under one cycle.  This probe asks whether that price is
// attached to the ISSUING wave (then a second, loader wave per SIMD could pay it instead) or to the CU (then nothing helps), and whether the
// gfx950 LDS-DMA form (`buffer_load_dwordx4 ... lds`: no destination registers) is any cheaper:
//   mode 0  4 computing waves (one per SIMD), v_mfma_f32_16x16x4_f32 stream, nothing else
//   mode 1  the same waves issue 4 x buffer_load_dwordx4 -> registers per 16 MFMAs (what k_loop_wino does)
//   mode 2  the same waves issue 4 x buffer_load_dwordx4 ... lds per 16 MFMAs and read the data back with 4 x ds_read_b128
//   mode 3  8 waves: the computing waves only read LDS (4 x ds_read_b128 per 16 MFMAs); a LOADER wave per SIMD issues the 4 loads -> registers
//   mode 4  8 waves: as 3, the loader waves use the LDS-DMA form
// Loader waves are paced with s_sleep so that they finish with the computing waves (their time is reported beside it).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_feeder_probe tools/mfma_feeder_probe.hip && /tmp/mfma_feeder_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr unsigned kStream = 2u << 20;      // bytes walked by the loads (k_loop_wino: 2 MiB per layer, the same addresses in every workgroup)

template <int MODE, int SLEEP>
__global__ __launch_bounds__(MODE >= 3 ? 512 : 256, 1) void probe(const float4* __restrict__ w, float* out, unsigned long long* cyc, int rounds) {
    __shared__ __attribute__((aligned(16))) float lds[8192];           // 32 KiB: [wave 4][stage 2][4 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wv >= 4;
    const int cw = wv & 3;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    const unsigned lbase = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float*)lds + (unsigned)cw * 8192u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(w), 0, kStream, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    f32x4 acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ld[8];
    for (int i = 0; i < 8; ++i) ld[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[4] = {0.5f + lane, 0.25f, 0.125f * lane, 1.f}, b = 0.001f * lane;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (!loader) {
        for (int it2 = 0; it2 < rounds; it2 += 2) {
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            const int it = it2 + par;
            const unsigned soff = ((unsigned)it * 16384u + (unsigned)cw * 4096u) & (kStream - 1);
            const unsigned st = lbase + (unsigned)par * 4096u;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a[m & 3]), "v"(b));
                if (m % 4 == 3) {
                    const int f = m / 4;
                    if (MODE == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:0" : "=v"(ld[par * 4 + f]) : "v"(voff + f * 1024u), "s"(rs), "s"(soff) : "memory");
                    if (MODE == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(st + f * 1024u), "v"(voff + f * 1024u), "s"(rs), "s"(soff) : "memory");
                    if (MODE >= 2) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[par * 4 + f]) : "v"(st + f * 1024u + voff) : "memory");
                }
            }
            if (MODE == 1 || MODE == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          }
        }
    } else {
        for (int it2 = 0; it2 < rounds; it2 += 2) {
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            const int it = it2 + par;
            const unsigned soff = ((unsigned)it * 16384u + (unsigned)cw * 4096u) & (kStream - 1);
            const unsigned st = lbase + (unsigned)par * 4096u;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (MODE == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:0" : "=v"(ld[par * 4 + f]) : "v"(voff + f * 1024u), "s"(rs), "s"(soff) : "memory");
                if (MODE == 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(st + f * 1024u), "v"(voff + f * 1024u), "s"(rs), "s"(soff) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_sleep(SLEEP);
          }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 4; ++r) sum += acc[m][r];
    for (int i = 0; i < 8; ++i) sum += ld[i][0] + ld[i][3];
    out[blockIdx.x * 512 + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

static float4* g_w; static float* g_out; static unsigned long long* g_cyc;

template <int MODE, int SLEEP>
int run(const char* what) {
    const int rounds = 2000, blocks = 256, threads = MODE >= 3 ? 512 : 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(g_cyc, 0, 4096 * 8 * 8));
    probe<MODE, SLEEP><<<blocks, threads>>>(g_w, g_out, g_cyc, rounds);
    CK(hipEventRecord(e0));
    probe<MODE, SLEEP><<<blocks, threads>>>(g_w, g_out, g_cyc, rounds);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 8);
    CK(hipMemcpy(h.data(), g_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double comp = 0, load = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? comp : load) += (double)h[b * 8 + w];
    comp /= blocks * 4; load /= blocks * 4;
    const double mf = (double)rounds * 16;
    printf("{\"mode\": %d, \"what\": \"%s\", \"loader_sleep\": %d, \"cycles_per_mfma\": %.2f, \"loader_cycles_per_16_mfma_period\": %.1f, \"tflops\": %.1f, \"kernel_ms\": %.3f}\n",
           MODE, what, SLEEP, comp / mf, load / rounds, blocks * 4 * mf * 2048.0 / (ms * 1e-3) / 1e12, ms);
    return 0;
}

int main() {
    CK(hipMalloc(&g_w, (size_t)kStream + 65536));
    CK(hipMemset(g_w, 0, (size_t)kStream + 65536));
    CK(hipMalloc(&g_out, 4096 * 512 * 4));
    CK(hipMalloc(&g_cyc, 4096 * 8 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0, 0>("4 computing waves, MFMA stream only")) return 1;
        if (run<1, 0>("computing waves: 4 x buffer_load_dwordx4 -> registers per 16 MFMAs")) return 1;
        if (run<2, 0>("computing waves: 4 x buffer_load_dwordx4 lds + 4 x ds_read_b128 per 16 MFMAs")) return 1;
        if (run<3, 4>("loader waves: 4 x buffer_load_dwordx4 -> registers; computing waves: 4 x ds_read_b128")) return 1;
        if (run<3, 6>("loader waves: 4 x buffer_load_dwordx4 -> registers; computing waves: 4 x ds_read_b128")) return 1;
        if (run<3, 7>("loader waves: 4 x buffer_load_dwordx4 -> registers; computing waves: 4 x ds_read_b128")) return 1;
        if (run<4, 4>("loader waves: 4 x buffer_load_dwordx4 lds; computing waves: 4 x ds_read_b128")) return 1;
        if (run<4, 6>("loader waves: 4 x buffer_load_dwordx4 lds; computing waves: 4 x ds_read_b128")) return 1;
        if (run<4, 7>("loader waves: 4 x buffer_load_dwordx4 lds; computing waves: 4 x ds_read_b128")) return 1;
    }
    return 0;
}
