// Developer probe (GPU): what does ONE instruction of another kind cost beside an fp32 MFMA stream, one wave per SIMD?
// The persistent loops are fp32-MFMA bound and every step carries loads, waits, scalar address arithmetic, LDS reads and a few vector adds;
// round 5's Winograd loop halves the MFMA time per instruction (v_mfma_f32_16x16x4_f32: 32 cycles) with the same filler count, and its steps
// measure 37 cycles per MFMA where k_loop's 64-cycle MFMAs measure 67.  This probe prices the fillers: a stream of MFMAs over four rotating
// accumulators (dependency distance 4 MFMAs) with NF fillers of ONE kind behind every MFMA, all 256 CUs, 4 waves per CU (one per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_filler_probe tools/mfma_filler_probe.hip && /tmp/mfma_filler_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { F_NONE = 0, F_SALU, F_WAIT, F_VALU, F_VMEM, F_DS, F_VMEM4, F_PK, F_ACCRD };

template <int TYPE>
__device__ __forceinline__ void filler(unsigned& sc, float& x, float y, f32x4 (&ld)[8], int i, const float4* gp, unsigned lp) {
    if (TYPE == F_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) : : "scc");
    if (TYPE == F_PK) { typedef float f32x2 __attribute__((ext_vector_type(2))); f32x2 p = {ld[0][0], ld[0][1]}, q = {y, y};
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q)); ld[0][0] = p[0]; ld[0][1] = p[1]; }
    if (TYPE == F_ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(y));
    if (TYPE == F_WAIT) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    if (TYPE == F_VALU) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (TYPE == F_VMEM) asm volatile("global_load_dword %0, %1, off" : "=v"(ld[i & 7][0]) : "v"(gp) : "memory");
    if (TYPE == F_VMEM4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[i & 7]) : "v"(gp) : "memory");
    if (TYPE == F_DS) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[i & 7]) : "v"(lp) : "memory");
}

// NF fillers behind every EVERY-th MFMA
template <int SHAPE, int TYPE, int NF, int EVERY = 1>      // SHAPE 16: v_mfma_f32_16x16x4_f32 (32 cycles), 32: v_mfma_f32_32x32x2_f32 (64 cycles)
__global__ __launch_bounds__(256, 1) void probe(const float4* __restrict__ w, float* out, unsigned long long* cyc, int rounds) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4096; i += 256) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc32[4];
    f32x4 acc16[4];
    for (int m = 0; m < 4; ++m) { for (int r = 0; r < 16; ++r) acc32[m][r] = 0.f; for (int r = 0; r < 4; ++r) acc16[m][r] = 0.f; }
    float a[4] = {0.5f + lane, 0.25f, 0.125f * lane, 1.f}, b = 0.001f * lane;
    unsigned sc = 0;
    float x = 1.f, y = 1e-9f;
    f32x4 ld[8];
    for (int i = 0; i < 8; ++i) ld[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* gp = w + (size_t)wv * 64 + lane;
    const unsigned lp = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float*)lds + (unsigned)lane * 16u;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (SHAPE == 16) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc16[m & 3]) : "v"(a[m & 3]), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc32[m & 3]) : "v"(a[m & 3]), "v"(b));
            if (m % EVERY == EVERY - 1) {
#pragma unroll
                for (int f = 0; f < NF; ++f) filler<TYPE>(sc, x, y, ld, m * NF + f, gp, lp);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = x + (float)sc;
    for (int m = 0; m < 4; ++m) { for (int r = 0; r < 16; ++r) sum += acc32[m][r]; for (int r = 0; r < 4; ++r) sum += acc16[m][r]; }
    for (int i = 0; i < 8; ++i) sum += ld[i][0] + ld[i][3];
    out[blockIdx.x * 256 + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
}

static float4* g_w; static float* g_out; static unsigned long long* g_cyc;

template <int SHAPE, int TYPE, int NF, int EVERY = 1>
int run(const char* tname) {
    const int rounds = 400, blocks = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<SHAPE, TYPE, NF, EVERY><<<blocks, 256>>>(g_w, g_out, g_cyc, rounds);
    CK(hipEventRecord(e0));
    probe<SHAPE, TYPE, NF, EVERY><<<blocks, 256>>>(g_w, g_out, g_cyc, rounds);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), g_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= h.size();
    const double mf = (double)rounds * 16;
    const double flop = SHAPE == 16 ? 2048.0 : 4096.0;
    printf("{\"mfma\": \"%s\", \"filler\": \"%s\", \"fillers\": %d, \"per_mfmas\": %d, \"cycles_per_mfma\": %.2f, \"tflops\": %.1f, \"ghz\": %.3f}\n",
           SHAPE == 16 ? "16x16x4_f32" : "32x32x2_f32", tname, NF, EVERY, mean / mf, blocks * 4 * mf * flop / (ms * 1e-3) / 1e12, mean / (ms * 1e-3) / 1e9);
    return 0;
}

template <int SHAPE, int TYPE>
int sweep(const char* tname) {
    if (run<SHAPE, TYPE, 1>(tname)) return 1;
    if (run<SHAPE, TYPE, 2>(tname)) return 1;
    if (run<SHAPE, TYPE, 3>(tname)) return 1;
    if (run<SHAPE, TYPE, 5>(tname)) return 1;
    return 0;
}

template <int SHAPE>
int all() {
    if (run<SHAPE, F_NONE, 0>("none")) return 1;
    if (sweep<SHAPE, F_SALU>("s_add_u32")) return 1;
    if (sweep<SHAPE, F_WAIT>("s_waitcnt (satisfied)")) return 1;
    if (sweep<SHAPE, F_VALU>("v_add_f32")) return 1;
    if (sweep<SHAPE, F_DS>("ds_read_b128")) return 1;
    if (sweep<SHAPE, F_PK>("v_pk_add_f32")) return 1;
    if (run<SHAPE, F_ACCRD, 1>("v_accvgpr_read_b32")) return 1;
    if (run<SHAPE, F_ACCRD, 2>("v_accvgpr_read_b32")) return 1;
    if (run<SHAPE, F_VMEM4, 1>("global_load_dwordx4 (1 KiB per wave, L2 hit)")) return 1;
    // the densities of the loops: 4 loads per 16 MFMAs, 8 vector adds per 64 MFMAs (one gap), 4 LDS reads per 64
    if (run<SHAPE, F_VMEM4, 1, 2>("global_load_dwordx4")) return 1;
    if (run<SHAPE, F_VMEM4, 1, 4>("global_load_dwordx4")) return 1;
    if (run<SHAPE, F_VMEM4, 2, 8>("global_load_dwordx4")) return 1;
    if (run<SHAPE, F_VMEM4, 4, 16>("global_load_dwordx4")) return 1;
    if (run<SHAPE, F_VALU, 1, 8>("v_add_f32")) return 1;
    if (run<SHAPE, F_VALU, 8, 16>("v_add_f32")) return 1;
    if (run<SHAPE, F_VALU, 2, 4>("v_add_f32")) return 1;
    if (run<SHAPE, F_DS, 4, 16>("ds_read_b128")) return 1;
    return 0;
}

int main() {
    CK(hipMalloc(&g_w, (size_t)1 << 20));
    CK(hipMemset(g_w, 0, (size_t)1 << 20));
    CK(hipMalloc(&g_out, 4096 * 512 * 4));
    CK(hipMalloc(&g_cyc, 4096 * 8 * 8));
    if (all<16>()) return 1;
    if (all<32>()) return 1;
    return 0;
}
