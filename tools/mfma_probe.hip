// Developer probe (GPU): cadence of v_mfma_f32_32x32x2_f32 with different filler instructions between them.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void probe(const float4* __restrict__ w, float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[256 * 48];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 256 * 48; i += 256) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float4* ap = w + (size_t)(blockIdx.x % 8) * 0 + wv * (96 * 256) + lane;   // all blocks stream the same 1.5 MB
    const float* bp = lds + (lane >> 5) * 4 * 48 + 8 + (lane & 31);
    float4 a[2][4];
    float b[2][4];
    for (int m = 0; m < 4; ++m) a[0][m] = ap[m * 64];
    for (int s = 0; s < 4; ++s) b[0][s] = bp[s * 48];
    SB();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kc = (it * 2 + u + 1) % 96;
            if (VARIANT == 1 || VARIANT == 3) {          // global loads of the next chunk
                for (int m = 0; m < 4; ++m) a[u ^ 1][m] = ap[(size_t)kc * 256 + m * 64];
            } else {
                for (int m = 0; m < 4; ++m) a[u ^ 1][m] = a[u][m];
            }
            if (VARIANT == 2 || VARIANT == 3) {          // LDS reads of the next chunk
                for (int s = 0; s < 4; ++s) b[u ^ 1][s] = bp[(kc & 31) * 8 * 48 + s * 48];
            } else {
                for (int s = 0; s < 4; ++s) b[u ^ 1][s] = b[u][s];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float av = s == 0 ? a[u][m].x : s == 1 ? a[u][m].y : s == 2 ? a[u][m].z : a[u][m].w;
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][s], acc[m], 0, 0, 0);
                }
            if (VARIANT == 1 || VARIANT == 3)
                for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            if (VARIANT == 2 || VARIANT == 3)
                for (int i = 0; i < 2; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            SB();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) sum += acc[m][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
}

template <int V>
void run(const char* name, const float4* w, float* out, unsigned long long* cyc, int blocks) {
    const int iters = 480;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<blocks, 256>>>(w, out, cyc, iters);
    hipEventRecord(e0);
    probe<V><<<blocks, 256>>>(w, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; unsigned long long mx = 0;
    for (auto v : h) { mean += (double)v; mx = v > mx ? v : mx; }
    mean /= h.size();
    const double mf = (double)iters * 2 * 16;
    printf("%-28s blocks=%4d  cycles/MFMA mean %.2f max %.2f   kernel %.3f ms -> %.1f TFLOP/s\n", name, blocks, mean / mf, mx / mf, ms,
           blocks * 4 * mf * 4096.0 / (ms * 1e-3) / 1e12);
}

int main() {
    float4* w; float* out; unsigned long long* cyc;
    hipMalloc(&w, (size_t)4 * 96 * 256 * 16 * 2);
    hipMemset(w, 0, (size_t)4 * 96 * 256 * 16 * 2);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&cyc, 4096 * 4 * 8);
    for (int blocks : {256, 512}) {
        run<0>("mfma only", w, out, cyc, blocks);
        run<1>("mfma + 4 global_load_x4", w, out, cyc, blocks);
        run<2>("mfma + ds_read", w, out, cyc, blocks);
        run<3>("mfma + global + ds", w, out, cyc, blocks);
    }
    return 0;
}
