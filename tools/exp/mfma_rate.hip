// Issue rate of the fp32 / bf16 matrix instructions on one wave per SIMD, with the shader clock measured beside the 100 MHz
// real-time counter.  Build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_rate.hip -o gpurun_out/mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t, int iters) {
    f32x4 acc[NACC];
    f32x16 acc32[NACC > 4 ? 4 : NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < (NACC > 4 ? 4 : NACC); ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(float)(threadIdx.x + i); bh[i] = (__bf16)(float)(threadIdx.x - i); }
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 64 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(rep + i) & 3], b[i & 3], acc[i], 0, 0, 0);
                if (KIND == 1) acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(rep + i) & 3], b[i & 3], acc32[i & 3], 0, 0, 0);
                if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
            }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < (NACC > 4 ? 4 : NACC); ++i) s += acc32[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { t[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = c1 - c0; t[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

template <int KIND, int NACC>
void run(const char* name, int grid, int iters) {
    float* out; unsigned long long* t;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&t, grid * 4 * 2 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, NACC>), dim3(grid), dim3(256), 0, 0, out, t, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), t, grid * 64, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < grid * 4; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
    c /= grid * 4; r /= grid * 4;
    const double n = 64.0 * iters;
    printf("%-34s grid %4d: %7.1f us (events)  memtime ticks/MFMA %6.2f  realtime(100MHz) ns/MFMA %6.2f  -> memtime rate %.0f MHz\n", name, grid, ms * 1e3, c / n,
           r * 10.0 / n, c / (r * 10.0) * 1e3);
    hipFree(out); hipFree(t);
}

int main() {
    for (int grid : {1, 256}) {
        run<0, 16>("f32 16x16x4, 16 accumulators", grid, 256);
        run<0, 4>("f32 16x16x4, 4 accumulators", grid, 256);
        run<0, 2>("f32 16x16x4, 2 accumulators", grid, 256);
        run<0, 1>("f32 16x16x4, 1 accumulator", grid, 256);
        run<1, 4>("f32 32x32x2, 4 accumulators", grid, 256);
        run<1, 1>("f32 32x32x2, 1 accumulator", grid, 256);
        run<2, 16>("bf16 16x16x32, 16 accumulators", grid, 256);
        run<2, 1>("bf16 16x16x32, 1 accumulator", grid, 256);
    }
    return 0;
}
