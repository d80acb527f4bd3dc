// Developer probe (GPU): achievable HBM bandwidth on this box - float4 grid-stride copy and read-only sum over buffers far
// larger than the 256 MiB Infinity Cache.  Recorded in profiles/roofline.json next to the fp32-MFMA issue-rate probe.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_probe.bin tools/hbm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
int main() {
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    float4 *a, *b; float* o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {2048, 4096, 8192, 16384}) {
        k_copy<<<blocks, 256>>>(a, b, n);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) k_copy<<<blocks, 256>>>(a, b, n);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy  4 GiB -> 4 GiB, %5d blocks: %.3f ms/launch, %.2f TB/s (read+write)\n", blocks, ms / 5, 2.0 * bytes * 5 / (ms * 1e-3) / 1e12);
        k_read<<<blocks, 256>>>(a, o, n);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) k_read<<<blocks, 256>>>(a, o, n);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("read  4 GiB,          %5d blocks: %.3f ms/launch, %.2f TB/s\n", blocks, ms / 5, 1.0 * bytes * 5 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
