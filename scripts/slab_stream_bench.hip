// Microbenchmark: can passes over the ROW-MAJOR observation matrix [N][F] f32 (32-feature slabs = one 128-byte line per row and
// block) stream at the HBM rate?  If yes, quantile selection and binning can work on the caller's matrix directly and the
// feature-major key copy (transpose: 2 streams of N F 4 bytes) disappears.
//   mode 0: read only (xor into a register)
//   mode 1: read + one LDS atomic per key on [32][1024] counters (bucket from cheap arithmetic)
//   mode 2: read + write 2-byte codes in the production layout [F/16][N][16] (8-byte stores: 4 codes of one row and group)
//   mode 3: mode 2 with a 2-probe LDS lookup per key
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/slab_stream_bench.hip -o scripts/bin/slab_stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <cstring>
static uint32_t *g_tab = nullptr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, int WAVES, int DEPTH>
__global__ __launch_bounds__(64 * WAVES) void k(const float *__restrict__ obs, int n, int F, int chunk_rows, uint16_t *__restrict__ codes,
                                                uint32_t *__restrict__ sink, const uint32_t *__restrict__ tab) {
    extern __shared__ uint32_t lds[];
    const int chunk = blockIdx.x, f0 = blockIdx.y * 32;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (MODE == 1 || MODE == 3) { for (int i = tid; i < 32 * 1024; i += 64 * WAVES) lds[i] = MODE == 3 ? i * 2654435761u : 0; __syncthreads(); }
    const int r_lo = chunk * chunk_rows, r_hi = min(n, r_lo + chunk_rows);
    const int lr = lane >> 3, q = lane & 7;
    constexpr int kStep = 8 * WAVES;
    auto load = [&](int r) -> float4 {
        // clamped, UNCONDITIONAL load: a load inside a branch gets an `s_waitcnt vmcnt(0)` right behind it and the prefetch is gone
        return *reinterpret_cast<const float4 *>(obs + static_cast<size_t>(min(r, n - 1)) * F + f0 + 4 * q);
    };
    int g = r_lo + wave * 8;
    float4 buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) buf[d] = load(g + d * kStep + lr);
    uint32_t acc = 0;
    uint32_t *cnt = lds + (4 * q) * 1024;
    // modes 4 / 5: lut [32][512] u32 (64 KiB); mode 4: cnt2 [32][512] packed u16 pairs (64 KiB); mode 5: bmp [32][32] x 8 B, slotf [32][260] u16, thr [32][256] u32
    uint32_t *lut = lds;
    uint32_t *cnt2 = lds + 32 * 513 + 32;
    uint32_t *bmp = lds + 32 * 513 + 32;
    uint16_t *slotf = reinterpret_cast<uint16_t *>(bmp + 32 * 66);
    uint32_t *thr = reinterpret_cast<uint32_t *>(slotf + 32 * 262);
    if (MODE == 4 || MODE == 5) {
        // synthetic but valid tables: 512 cells, the populated binades get buckets in proportion to a normal population
        for (int i = tid; i < 32 * 513; i += 64 * WAVES) lut[i] = tab[(i % 513) & 511];
        if (MODE == 4) for (int i = tid; i < 32 * 513; i += 64 * WAVES) cnt2[i] = 0;
        if (MODE == 5) {
            for (int i = tid; i < 32 * 33; i += 64 * WAVES) { bmp[2 * i] = 0x11111111u; bmp[2 * i + 1] = (i % 33) * 8; }
            for (int i = tid; i < 32 * 262; i += 64 * WAVES) slotf[i] = (i % 262) & 255;
            for (int i = tid; i < 32 * 257; i += 64 * WAVES) thr[i] = 0x80000000u + i * 7919u;
        }
        __syncthreads();
    }
    const int grp = q >> 2, sub = q & 3;
    for (; g < r_hi; g += kStep) {
        const float4 c = buf[0];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) buf[d] = buf[d + 1];
        buf[DEPTH - 1] = load(g + DEPTH * kStep + lr);
        const uint32_t k0 = __float_as_uint(c.x), k1 = __float_as_uint(c.y), k2 = __float_as_uint(c.z), k3 = __float_as_uint(c.w);
        if (MODE == 0) acc ^= k0 ^ k1 ^ k2 ^ k3;
        if (MODE == 4 && g + lr < r_hi) {
            // realistic count pass: order-preserving key, cell LUT probe (top 9 bits -> base | scale), bucket = base + (low23 * scale >> 23),
            // uint16-pair packed counters [32][1024]
            const uint32_t kk[4] = {k0, k1, k2, k3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t u = kk[j];
                if ((u << 1) == 0) u = 0;
                uint32_t key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                if ((u & 0x7fffffffu) > 0x7f800000u) key = 0;
                const uint32_t e = lut[(4 * q + j) * 513 + (key >> 23)];
                const uint32_t bucket = (e >> 16) + (__umul24((key >> 7) & 0xffffu, e & 0xffffu) >> 16);
                atomicAdd(cnt2 + (4 * q + j) * 513 + (bucket >> 1), 1u << ((bucket & 1) * 16));
            }
        }
        if (MODE == 5 && g + lr < r_hi) {
            // realistic binning pass: cell LUT probe, bitmap+prefix probe (8 bytes), slot probe, ~12 % of the keys one threshold compare
            const uint32_t kk[4] = {k0, k1, k2, k3};
            uint32_t cc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t u = kk[j];
                if ((u << 1) == 0) u = 0;
                uint32_t key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                if ((u & 0x7fffffffu) > 0x7f800000u) key = 0;
                const int f = 4 * q + j;
                const uint32_t e = lut[f * 513 + (key >> 23)];
                const uint32_t bucket = (e >> 16) + (__umul24((key >> 7) & 0xffffu, e & 0xffffu) >> 16);
                const uint2 bm = *reinterpret_cast<const uint2 *>(bmp + f * 66 + (bucket >> 5) * 2);
                const uint32_t below = bm.y + __popc(bm.x & ((1u << (bucket & 31)) - 1u));
                uint32_t code = slotf[f * 262 + below];
                if ((bm.x >> (bucket & 31)) & 1u) code += thr[f * 257 + (code & 255)] < key ? 1u : 0u;
                cc[j] = code;
            }
            uint2 o = make_uint2(cc[0] | (cc[1] << 16), cc[2] | (cc[3] << 16));
            const size_t gi = static_cast<size_t>(blockIdx.y * 2 + grp);
            *reinterpret_cast<uint2 *>(codes + (gi * n + g + lr) * 16 + sub * 4) = o;
        }
        if (MODE == 1 && g + lr < r_hi) {
            atomicAdd(cnt + ((k0 >> 13) & 1023), 1u); atomicAdd(cnt + 1024 + ((k1 >> 13) & 1023), 1u);
            atomicAdd(cnt + 2048 + ((k2 >> 13) & 1023), 1u); atomicAdd(cnt + 3072 + ((k3 >> 13) & 1023), 1u);
        }
        if ((MODE == 2 || MODE == 3) && g + lr < r_hi) {
            uint32_t c0 = (k0 >> 15) & 255, c1 = (k1 >> 15) & 255, c2 = (k2 >> 15) & 255, c3 = (k3 >> 15) & 255;
            if (MODE == 3) {
                c0 = cnt[(k0 >> 13) & 1023]; c1 = cnt[1024 + ((k1 >> 13) & 1023)]; c2 = cnt[2048 + ((k2 >> 13) & 1023)]; c3 = cnt[3072 + ((k3 >> 13) & 1023)];
                c0 = cnt[c0 & 1023] & 255; c1 = cnt[1024 + (c1 & 1023)] & 255; c2 = cnt[2048 + (c2 & 1023)] & 255; c3 = cnt[3072 + (c3 & 1023)] & 255;
            }
            uint2 o = make_uint2(c0 | (c1 << 16), c2 | (c3 << 16));
            const size_t gi = static_cast<size_t>(blockIdx.y * 2 + grp);
            *reinterpret_cast<uint2 *>(codes + (gi * n + g + lr) * 16 + sub * 4) = o;
        }
    }
    if (MODE == 1 || MODE == 4) { __syncthreads(); for (int i = tid; i < 32 * 1024; i += 64 * WAVES) acc ^= lds[i]; }

    if (acc == 0x1234567u) sink[0] = acc;
}

template <int WAVES, int DEPTH>
__global__ __launch_bounds__(64 * WAVES) void k16(const float *__restrict__ obs, int n, int F, int chunk_rows, uint32_t *__restrict__ sink) {
    const int chunk = blockIdx.x, f0 = blockIdx.y * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r_lo = chunk * chunk_rows, r_hi = min(n, r_lo + chunk_rows);
    const int lr = lane >> 2, q = lane & 3;
    constexpr int kStep = 16 * WAVES;
    auto load = [&](int r) -> float4 { return *reinterpret_cast<const float4 *>(obs + static_cast<size_t>(min(r, n - 1)) * F + f0 + 4 * q); };
    int g = r_lo + wave * 16;
    float4 buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) buf[d] = load(g + d * kStep + lr);
    uint32_t acc = 0;
    for (; g < r_hi; g += kStep) {
        const float4 c = buf[0];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) buf[d] = buf[d + 1];
        buf[DEPTH - 1] = load(g + DEPTH * kStep + lr);
        acc ^= __float_as_uint(c.x) ^ __float_as_uint(c.y) ^ __float_as_uint(c.z) ^ __float_as_uint(c.w);
    }
    if (acc == 0x1234567u) sink[0] = acc;
}
template <int WAVES, int DEPTH>
int run16(const float *dx, int N, int F, uint32_t *ds, int chunk_rows, const char *what) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k16<WAVES, DEPTH>), dim3((N + chunk_rows - 1) / chunk_rows, F / 16), dim3(64 * WAVES), 0, 0, dx, N, F, chunk_rows, ds);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("%-44s waves %2d depth %d chunk %6d: %7.1f us  read %.2f TB/s\n", what, WAVES, DEPTH, chunk_rows, ms * 1e3, double(N) * F * 4 / ms / 1e9);
    return 0;
}
template <int MODE, int WAVES, int DEPTH>
int run(const float *dx, int N, int F, uint16_t *dc, uint32_t *ds, int chunk_rows, const char *what) {
    const size_t lds = (MODE == 1 || MODE == 3) ? 32 * 1024 * 4 : (MODE == 4 ? (32 * 513 * 2 + 64) * 4 : (MODE == 5 ? (32 * 513 + 32 + 32 * 66 + 32 * 131 + 32 * 257 + 64) * 4 : 0));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, WAVES, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<MODE, WAVES, DEPTH>), dim3((N + chunk_rows - 1) / chunk_rows, F / 32), dim3(64 * WAVES), lds, 0, dx, N, F, chunk_rows, dc, ds, g_tab);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, a, b));
    }
    const double rd = double(N) * F * 4, wr = MODE >= 2 ? double(N) * F * 2 : 0;
    printf("%-44s waves %2d depth %d chunk %6d: %7.1f us  read %.2f TB/s  (read+write %.2f TB/s)\n", what, WAVES, DEPTH, chunk_rows, ms * 1e3, rd / ms / 1e9, (rd + wr) / ms / 1e9);
    return 0;
}
int main() {
    const int N = 1 << 20, F = 128;
    float *dx; uint16_t *dc; uint32_t *ds;
    CK(hipMalloc(&dx, size_t(N) * F * 4)); CK(hipMalloc(&dc, size_t(N) * F * 2)); CK(hipMalloc(&ds, 64));
    {   // standard normal data; cell table: buckets per cell in proportion to the normal mass of the cell (1024 buckets)
        std::vector<float> hx(size_t(N) * F);
        std::mt19937 rng(3); std::normal_distribution<float> nd;
        for (auto &v : hx) v = nd(rng);
        CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        std::vector<double> mass(512, 0.0);
        for (size_t i = 0; i < hx.size(); i += 97) {
            uint32_t u; memcpy(&u, &hx[i], 4);
            const uint32_t key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            mass[key >> 23] += 1.0;
        }
        double tot = 0; for (double m : mass) tot += m;
        std::vector<uint32_t> tab(512);
        uint32_t base = 0;
        for (int c = 0; c < 512; ++c) {
            uint32_t sc = uint32_t(mass[c] / tot * 1000.0);
            if (base + sc > 1023) sc = 1023 - base;
            tab[c] = (base << 16) | sc;
            base += sc;
        }
        CK(hipMalloc(&g_tab, 512 * 4));
        CK(hipMemcpy(g_tab, tab.data(), 512 * 4, hipMemcpyHostToDevice));
    }
    if (run16<16, 2>(dx, N, F, ds, 32768, "16-feature slabs (64-B pieces), read only")) return 1;
    if (run16<16, 4>(dx, N, F, ds, 32768, "16-feature slabs (64-B pieces), read only")) return 1;
    if (run16<16, 4>(dx, N, F, ds, 8192, "16-feature slabs (64-B pieces), read only")) return 1;
    if (run<0, 16, 2>(dx, N, F, dc, ds, 16384, "mode 0 read only")) return 1;
    if (run<0, 16, 4>(dx, N, F, dc, ds, 16384, "mode 0 read only")) return 1;
    if (run<0, 16, 8>(dx, N, F, dc, ds, 16384, "mode 0 read only")) return 1;
    if (run<0, 8, 8>(dx, N, F, dc, ds, 16384, "mode 0 read only")) return 1;
    if (run<0, 16, 4>(dx, N, F, dc, ds, 4096, "mode 0 read only (1024 blocks)")) return 1;
    if (run<0, 4, 4>(dx, N, F, dc, ds, 1024, "mode 0 read only (4096 small blocks)")) return 1;
    if (run<1, 16, 4>(dx, N, F, dc, ds, 16384, "mode 1 read + LDS count")) return 1;
    if (run<1, 16, 8>(dx, N, F, dc, ds, 16384, "mode 1 read + LDS count")) return 1;
    if (run<2, 16, 4>(dx, N, F, dc, ds, 16384, "mode 2 read + code write")) return 1;
    if (run<2, 16, 8>(dx, N, F, dc, ds, 16384, "mode 2 read + code write")) return 1;
    if (run<2, 4, 4>(dx, N, F, dc, ds, 1024, "mode 2 read + code write (small blocks)")) return 1;
    if (run<3, 16, 4>(dx, N, F, dc, ds, 16384, "mode 3 read + 2 LDS probes + code write")) return 1;
    if (run<3, 16, 8>(dx, N, F, dc, ds, 16384, "mode 3 read + 2 LDS probes + code write")) return 1;
    if (run<4, 16, 2>(dx, N, F, dc, ds, 16384, "mode 4 realistic count pass")) return 1;
    if (run<4, 16, 4>(dx, N, F, dc, ds, 16384, "mode 4 realistic count pass")) return 1;
    if (run<5, 16, 2>(dx, N, F, dc, ds, 16384, "mode 5 realistic binning pass")) return 1;
    if (run<5, 16, 4>(dx, N, F, dc, ds, 16384, "mode 5 realistic binning pass")) return 1;
    if (run<5, 16, 4>(dx, N, F, dc, ds, 4096, "mode 5 realistic binning pass")) return 1;
    return 0;
}
