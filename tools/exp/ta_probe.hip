// Gather-throughput probe (round 3): how fast can a CU pull 16-byte quads of gathered rows out of L2, as a function of the
// lane -> address mapping of the wave's load instruction?  Same shapes as the dec0 forward layer (rows of C floats, table of
// neighbours, 16 batch entries per vertex contiguous), no arithmetic beyond keeping the loads alive.
//   mode 0  "mfma":      lane (r = lane & 15, kq = lane >> 4) loads quad (16 ks + 4 kq) of row r       (direct kernel)
//   mode 1  "mfma32":    lane (r, kq) loads the two adjacent quads (8 kq, 8 kq + 4) of row r              (split3 kernel)
//   mode 2  "coalesced": lane l loads quad (l & 7) of row (l >> 3) + 8 i                                  (8 full lines / instruction)
// build: hipcc --offload-arch=gfx950 -O3 -o ta_probe ta_probe.hip ; run: ./ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ x, const int* __restrict__ table, float* out, int R, int S, int C,
                                             int B, int n_vt, int xcd_local) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-contiguous tile order (as sh_xcd_remap in the library): every XCD sweeps a contiguous range of tiles, so the S-fold
    // re-reads of a neighbour row hit in that XCD's 4 MiB L2; xcd_local == 0 deals tiles round-robin (rows come from the fabric)
    int tile = blockIdx.x;
    if (xcd_local) {
        const int nwg = gridDim.x, xcd = tile & 7, local = tile >> 3, qq = nwg >> 3, rr = nwg & 7;
        tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + local;
    }
    const int bt = tile / n_vt, vt = tile - bt * n_vt;
    const int v = min(vt * 4 + wave, R - 1), b0 = bt * 16;
    const int* tl = table + (long)v * S;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nchunk = S * C / 32;
    // per chunk a wave fetches 16 rows (batch entries b0..b0+15 of one neighbour) x 32 floats = 2 KiB in two instructions
#pragma unroll 4
    for (int c = 0; c < nchunk; ++c) {
        const int k = 32 * c, s = k / C, ch = k - s * C;
        const float* base = x + ((long)tl[s] * B + b0) * C + ch;
        f32x4 a, b;
        if (MODE == 0) {
            const int r = lane & 15, kq = lane >> 4;
            a = *reinterpret_cast<const f32x4*>(base + (long)r * C + 4 * kq);
            b = *reinterpret_cast<const f32x4*>(base + (long)r * C + 16 + 4 * kq);
        } else if (MODE == 1) {
            const int r = lane & 15, kq = lane >> 4;
            a = *reinterpret_cast<const f32x4*>(base + (long)r * C + 8 * kq);
            b = *reinterpret_cast<const f32x4*>(base + (long)r * C + 8 * kq + 4);
        } else {
            const int r = lane >> 3, q = lane & 7;
            a = *reinterpret_cast<const f32x4*>(base + (long)r * C + 4 * q);
            b = *reinterpret_cast<const f32x4*>(base + (long)(r + 8) * C + 4 * q);
        }
        acc += a; acc += b;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[0] = 1.f;
}

int main() {
    struct Cfg { int R, S, C; } cfgs[] = {{863, 8, 128}, {1724, 8, 64}, {3446, 11, 32}, {6891, 10, 32}};
    const int B = 64;
    for (auto cf : cfgs) {
        const int R = cf.R, S = cf.S, C = cf.C;
        std::vector<int> tab((size_t)R * S);
        srand(1);
        for (int v = 0; v < R; ++v)
            for (int s = 0; s < S; ++s) { int n = v + (rand() % 41) - 20; tab[(size_t)v * S + s] = n < 0 ? 0 : (n >= R ? R - 1 : n); }   // mesh-like locality
        float* x; int* t; float* out;
        hipMalloc(&x, (size_t)R * B * C * 4); hipMalloc(&t, tab.size() * 4); hipMalloc(&out, 4);
        hipMemset(x, 0, (size_t)R * B * C * 4);
        hipMemcpy(t, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
        const int n_vt = (R + 3) / 4, grid = n_vt * (B / 16);
        const double bytes = (double)grid * 4 * (S * C / 32) * 2048.0;
        for (int xl = 0; xl < 2; ++xl)
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, x, t, out, R, S, C, B, n_vt, xl);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, x, t, out, R, S, C, B, n_vt, xl);
                else hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, x, t, out, R, S, C, B, n_vt, xl);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it > 0 && ms < best) best = ms;
            }
            printf("R=%d S=%d C=%d grid=%d xcd_local=%d mode=%d: %.1f us, %.2f TB/s gathered (%.0f MB)\n", R, S, C, grid, xl, mode, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / 1e6);
        }
        hipFree(x); hipFree(t); hipFree(out);
    }
    return 0;
}
