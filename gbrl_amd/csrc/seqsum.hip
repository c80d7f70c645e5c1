// seqsum.hip -- the value of a SEQUENTIAL float32 sum, s = fl(...fl(fl(x0 + x1) + x2)... + x[M-1]), evaluated in parallel and bit for bit.
//
// Why: the near-tie replay (neartie.hip) has to reproduce the reference's float32 sums in row order (node.cpp:336-352), and for a node of 2^20
// rows such a chain is 10^6 dependent adds -- milliseconds on one lane, with the whole GPU idle.  The sum is not associative, but its rounding
// is simple between two changes of the running sum's exponent: while |s| stays inside one binade [2^e, 2^(e+1)) every partial sum is an
// integer multiple A of u = 2^(e-23), and adding x rounds the exact A + x/u to the nearest integer, ties to even:
//         A' = A + floor(x/u) + [frac(x/u) > 1/2] + [frac(x/u) == 1/2] * ((A + floor(x/u)) odd)
// -- a function of A that depends on A only through its PARITY.  Such functions compose associatively: a run of elements is summarised, for
// each starting parity p, by (delta_p, min_p, max_p) = the total change of A and the extremes of the partial sums relative to the start.
//   1. k_seq_blocksum  : fp64 sum of every block of 256 elements                      (all CUs)
//   2. k_seq_prefix    : per chain, exclusive fp64 prefix over its blocks -> the running sum every block starts from, to ~1e-16
//   3. k_seq_summary   : per block, the summary under the exponent that prefix predicts (all CUs)
//   4. k_seq_stitch    : one wave per chain walks the summaries in order: when the TRUE running sum has the predicted exponent and
//                        A0 + min_p .. A0 + max_p stay inside the binade (same sign), the block is applied in O(1); otherwise the block's 256
//                        elements are added one by one (the start of a chain, a crossing of a power of two, a cancellation).
// The result never depends on a prediction being right -- a wrong one only costs the serial fallback for that block.
// Prototype with the same arithmetic: scripts/experiments/seqsum_prototype.py (300 random chains against the plain loop, ties included).
#include "kernels.h"
#include "kernels_common.h"

#include <algorithm>
#include <vector>

#pragma clang fp contract(off)

namespace gbrl {
namespace kern {

namespace {

constexpr int kSeqBlock = 256;            // elements per summary (one wave, four per lane)
constexpr int kSeqBig = 1 << 28;          // clamp of a summary's fields (anything beyond 2^25 already fails the binade check; two clamped values add without overflow)

struct SeqSumm { int d[2], lo[2], hi[2]; };   // per starting parity: total change, least and greatest partial sum (relative to the start, after >= 1 element)

__device__ __forceinline__ int seq_clamp(int v) { return max(-kSeqBig, min(kSeqBig, v)); }

// one element under ulp exponent e (u = 2^(e - 23)): f = floor(x / u), h = 0 (fraction below a half) | 1 (above) | 2 (tie); false: not summarisable.
// An element whose exponent reaches the running sum's (k <= 0) always takes the sum out of its binade (same sign: beyond 2^(e+1); opposite: below
// 2^e or through zero), so it is not summarisable by definition -- which keeps every quantity inside 32 bits (|f| < 2^23).
__device__ __forceinline__ bool seq_element(float x, int e, int &f, int &h) {
    const uint32_t b = __float_as_uint(x);
    const int ex = static_cast<int>((b >> 23) & 0xffu);
    if (ex == 0xff) return false;                               // inf / nan: the serial loop decides
    const int m = ex ? static_cast<int>((b & 0x7fffffu) | 0x800000u) : static_cast<int>(b & 0x7fffffu);
    const int sm = (b >> 31) ? -m : m;                          // x = sm * 2^(ee - 23)
    const int ee = ex ? ex - 127 : -126;
    const int k = e - ee;
    if (k <= 0) return false;
    if (k >= 25) {                                              // |x| < u / 2: positive rounds away, negative floors to -1 and rounds back up
        f = sm < 0 ? -1 : 0; h = sm < 0 ? 1 : 0;
        return true;
    }
    f = sm >> k;                                                // arithmetic shift = floor
    const int rem = sm - (f << k), half = 1 << (k - 1);
    h = rem < half ? 0 : (rem > half ? 1 : 2);
    return true;
}
__device__ __forceinline__ SeqSumm seq_one(int f, int h) {
    SeqSumm s;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        int t = f;
        if (h == 1) t += 1;
        else if (h == 2) t += (p + f) & 1;
        s.d[p] = t; s.lo[p] = t; s.hi[p] = t;
    }
    return s;
}
// a, then b
__device__ __forceinline__ SeqSumm seq_compose(const SeqSumm &a, const SeqSumm &b) {
    SeqSumm r;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int d = a.d[p];
        const int q = (p + d) & 1;
        r.d[p] = seq_clamp(d + b.d[q]);
        r.lo[p] = seq_clamp(min(a.lo[p], d + b.lo[q]));
        r.hi[p] = seq_clamp(max(a.hi[p], d + b.hi[q]));
    }
    return r;
}
__device__ __forceinline__ long long seq_shfl_xor(long long v, int o) {
    const int lo = __shfl_xor(static_cast<int>(v & 0xffffffffll), o, kWave), hi = __shfl_xor(static_cast<int>(v >> 32), o, kWave);
    return (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo);
}
__device__ __forceinline__ double seq_shfl_xor_d(double v, int o) { return __longlong_as_double(seq_shfl_xor(__double_as_longlong(v), o)); }

// flat block b -> its chain (the chains' first blocks are ascending)
__device__ __forceinline__ int seq_chain_of(const SeqChain *__restrict__ chains, int n_chains, uint32_t b) {
    int lo = 0, hi = n_chains - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (chains[mid].blk0 <= b) lo = mid; else hi = mid - 1; }
    return lo;
}
// the block's elements, four consecutive ones per lane (zeros beyond the chain's end: x + 0 = x)
__device__ __forceinline__ void seq_load(const SeqChain &c, uint32_t lb, int lane, float (&v)[4]) {
    const uint32_t i0 = lb * kSeqBlock + 4u * lane;
    if (i0 + 4 <= c.len && (reinterpret_cast<uintptr_t>(c.x + i0) & 15) == 0) {
        const float4 t = *reinterpret_cast<const float4 *>(c.x + i0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u < c.len ? c.x[i0 + u] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void k_seq_blocksum(const SeqChain *__restrict__ chains, int n_chains, uint32_t n_blocks, double *__restrict__ blk) {
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const int lane = threadIdx.x & 63;
    const int ci = seq_chain_of(chains, n_chains, b);
    const SeqChain c = chains[ci];
    if (b - c.blk0 >= (c.len + kSeqBlock - 1) / kSeqBlock) return;   // (n_blocks may be an upper bound: nothing behind the last chain's blocks)
    float v[4];
    seq_load(c, b - c.blk0, lane, v);
    double s = ((static_cast<double>(v[0]) + static_cast<double>(v[1])) + static_cast<double>(v[2])) + static_cast<double>(v[3]);
    for (int o = 1; o < kWave; o <<= 1) s += seq_shfl_xor_d(s, o);
    if (lane == 0) blk[b] = s;
}

// one wave per chain: blk[b] <- sum of the chain's blocks before b
__global__ __launch_bounds__(64) void k_seq_prefix(const SeqChain *__restrict__ chains, int n_chains, double *__restrict__ blk) {
    const int ci = blockIdx.x, lane = threadIdx.x;
    if (ci >= n_chains) return;
    const SeqChain c = chains[ci];
    const uint32_t nb = (c.len + kSeqBlock - 1) / kSeqBlock;
    double run = 0.0;
    for (uint32_t b0 = 0; b0 < nb; b0 += kWave) {
        const uint32_t b = b0 + lane;
        const double mine = b < nb ? blk[c.blk0 + b] : 0.0;
        double incl = mine;
        for (int o = 1; o < kWave; o <<= 1) {
            const double up = __longlong_as_double((static_cast<long long>(__shfl_up(static_cast<int>(__double_as_longlong(incl) >> 32), o, kWave)) << 32) |
                                                   static_cast<unsigned int>(__shfl_up(static_cast<int>(__double_as_longlong(incl) & 0xffffffffll), o, kWave)));
            if (lane >= o) incl += up;
        }
        if (b < nb) blk[c.blk0 + b] = run + (incl - mine);
        const long long tot = __double_as_longlong(incl);
        run += __longlong_as_double((static_cast<long long>(__shfl(static_cast<int>(tot >> 32), kWave - 1, kWave)) << 32) |
                                    static_cast<unsigned int>(__shfl(static_cast<int>(tot & 0xffffffffll), kWave - 1, kWave)));
    }
}

// summaries as six planes + the exponent they were formed under (-1000: not summarisable)
__global__ __launch_bounds__(256) void k_seq_summary(const SeqChain *__restrict__ chains, int n_chains, uint32_t n_blocks, const double *__restrict__ blk,
                                                     int32_t *__restrict__ planes /*[6][n_blocks]*/, int32_t *__restrict__ expo) {
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const int lane = threadIdx.x & 63;
    const int ci = seq_chain_of(chains, n_chains, b);
    const SeqChain c = chains[ci];
    if (b - c.blk0 >= (c.len + kSeqBlock - 1) / kSeqBlock) return;
    const float start = static_cast<float>(static_cast<double>(c.start) + blk[b]);   // the running sum this block will (very nearly) start from
    const uint32_t sb = __float_as_uint(start);
    const int sex = static_cast<int>((sb >> 23) & 0xffu);
    int e = -1000;
    if (sex > 20 && sex < 235) e = sex - 127;                     // normal, away from the ends of the range
    float v[4];
    seq_load(c, b - c.blk0, lane, v);
    bool ok = e != -1000;
    SeqSumm s{};
    if (ok) {
        int f, h;
        ok = seq_element(v[0], e, f, h);
        if (ok) s = seq_one(seq_clamp(f), h);
#pragma unroll
        for (int u = 1; u < 4; ++u) {
            if (!ok) break;
            ok = seq_element(v[u], e, f, h);
            if (ok) s = seq_compose(s, seq_one(seq_clamp(f), h));
        }
    }
    if (__ballot(!ok) != 0ull) { if (lane == 0) expo[b] = -1000; return; }
    for (int o = 1; o < kWave; o <<= 1) {
        SeqSumm other;
#pragma unroll
        for (int p = 0; p < 2; ++p) { other.d[p] = __shfl_xor(s.d[p], o, kWave); other.lo[p] = __shfl_xor(s.lo[p], o, kWave); other.hi[p] = __shfl_xor(s.hi[p], o, kWave); }
        s = (lane & o) ? seq_compose(other, s) : seq_compose(s, other);     // the lower lanes' elements come first
    }
    if (lane == 0) {
        expo[b] = e;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            planes[(0 + p) * static_cast<size_t>(n_blocks) + b] = s.d[p];
            planes[(2 + p) * static_cast<size_t>(n_blocks) + b] = s.lo[p];
            planes[(4 + p) * static_cast<size_t>(n_blocks) + b] = s.hi[p];
        }
    }
}

// Does the running sum s (normal, exponent e) stay inside its binade through a run summarised by (d, lo, hi)[parity]?  If so apply it.
__device__ __forceinline__ bool seq_apply(float &s, int e, int d0, int d1, int lo0, int lo1, int hi0, int hi1) {
    const uint32_t sb = __float_as_uint(s);
    if (static_cast<int>((sb >> 23) & 0xffu) - 127 != e) return false;
    const int m = static_cast<int>((sb & 0x7fffffu) | 0x800000u);
    const int A0 = (sb >> 31) ? -m : m;
    const bool odd = (A0 & 1) != 0;
    const int d = odd ? d1 : d0, lo = odd ? lo1 : lo0, hi = odd ? hi1 : hi0;
    const bool ok = A0 > 0 ? (A0 + lo >= (1 << 23) && A0 + hi < (1 << 24)) : (A0 + hi <= -(1 << 23) && A0 + lo > -(1 << 24));
    if (ok) s = ldexpf(static_cast<float>(A0 + d), e - 23);     // exact: |A0 + d| < 2^24
    return ok;
}

// one wave per chain; every lane carries the same running sum (the control flow is wave-uniform).  Two levels: the summaries of 16 consecutive
// blocks formed under ONE exponent are composed (they are the same kind of function) and tried first -- a chain that drifts away from zero (a
// dot chain, a column with a mean) crosses a power of two a few dozen times in 2^20 elements and takes 4096 elements per step in between; a
// group that fails is walked block by block, a block that fails element by element.
constexpr int kSeqGroup = 16;
__global__ __launch_bounds__(64) void k_seq_stitch(const SeqChain *__restrict__ chains, int n_chains, uint32_t n_blocks, const int32_t *__restrict__ planes,
                                                   const int32_t *__restrict__ expo, float *__restrict__ out, uint32_t *__restrict__ n_slow /*nullable: blocks added one by one*/) {
    const int ci = blockIdx.x, lane = threadIdx.x;
    if (ci >= n_chains) return;
    const SeqChain c = chains[ci];
    const uint32_t nb = (c.len + kSeqBlock - 1) / kSeqBlock;
    float s = c.start;
    uint32_t slow = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += kWave) {
        const uint32_t b = c.blk0 + b0 + lane;
        const bool in = b0 + lane < nb;
        const int cnt = static_cast<int>(min(static_cast<uint32_t>(kWave), nb - b0));
        int my_e = in ? expo[b] : -1000;
        SeqSumm mine;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            mine.d[p] = in ? planes[(0 + p) * static_cast<size_t>(n_blocks) + b] : 0;
            mine.lo[p] = in ? planes[(2 + p) * static_cast<size_t>(n_blocks) + b] : kSeqBig;       // (beyond the chain's end: the identity)
            mine.hi[p] = in ? planes[(4 + p) * static_cast<size_t>(n_blocks) + b] : -kSeqBig;
        }
        if (!in) my_e = __builtin_amdgcn_readlane(my_e, cnt - 1);       // the padding takes the last block's exponent: it must not break a group
        // the groups' summaries: an ordered composition inside every 16-lane segment; valid when the 16 exponents agree
        SeqSumm grp = mine;
        int g_ok = my_e != -1000 ? 1 : 0;
        for (int o = 1; o < kSeqGroup; o <<= 1) {
            SeqSumm other;
#pragma unroll
            for (int p = 0; p < 2; ++p) { other.d[p] = __shfl_xor(grp.d[p], o, kWave); other.lo[p] = __shfl_xor(grp.lo[p], o, kWave); other.hi[p] = __shfl_xor(grp.hi[p], o, kWave); }
            const int oe = __shfl_xor(my_e, o, kWave), ook = __shfl_xor(g_ok, o, kWave);
            g_ok = g_ok & ook & (oe == my_e ? 1 : 0);
            grp = (lane & o) ? seq_compose(other, grp) : seq_compose(grp, other);
        }
        for (int g0 = 0; g0 < cnt; g0 += kSeqGroup) {
            const int ge = __builtin_amdgcn_readlane(my_e, g0);
            if (__builtin_amdgcn_readlane(g_ok, g0) &&
                seq_apply(s, ge, __builtin_amdgcn_readlane(grp.d[0], g0), __builtin_amdgcn_readlane(grp.d[1], g0), __builtin_amdgcn_readlane(grp.lo[0], g0),
                          __builtin_amdgcn_readlane(grp.lo[1], g0), __builtin_amdgcn_readlane(grp.hi[0], g0), __builtin_amdgcn_readlane(grp.hi[1], g0)))
                continue;
            const int g1 = min(cnt, g0 + kSeqGroup);
            float v[4], vn[4];
            seq_load(c, b0 + g0, lane, v);           // the elements of the block being tried, and of the next one, travel while the summaries are tried
            for (int j = g0; j < g1; ++j) {
                if (j + 1 < g1) seq_load(c, b0 + j + 1, lane, vn);
                const int e = __builtin_amdgcn_readlane(my_e, j);
                const bool fast = e != -1000 && seq_apply(s, e, __builtin_amdgcn_readlane(mine.d[0], j), __builtin_amdgcn_readlane(mine.d[1], j), __builtin_amdgcn_readlane(mine.lo[0], j),
                                                          __builtin_amdgcn_readlane(mine.lo[1], j), __builtin_amdgcn_readlane(mine.hi[0], j), __builtin_amdgcn_readlane(mine.hi[1], j));
                if (!fast) {
                    ++slow;
                    // sixteen elements (four lanes' worth) are read into scalars before their adds: the broadcasts do not depend on the chain
                    for (int l = 0; l < kWave; l += 4) {
                        float x[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int u = 0; u < 4; ++u) x[q * 4 + u] = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v[u])), l + q)));
#pragma unroll
                        for (int q = 0; q < 16; ++q) s = s + x[q];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = vn[u];
            }
        }
    }
    if (lane == 0) { out[ci] = s; if (n_slow) atomicAdd(n_slow, slow); }
}

}  // namespace

size_t seq_sums_scratch_bytes(uint32_t n_blocks) { return static_cast<size_t>(n_blocks) * (sizeof(double) + 7 * sizeof(int32_t)) + 256; }

void seq_sums(const SeqChain *d_chains, int n_chains, uint32_t n_blocks, void *d_scratch, float *d_out, uint32_t *d_n_slow, hipStream_t s) {
    if (n_chains <= 0) return;
    char *base = static_cast<char *>(d_scratch);
    double *blk = reinterpret_cast<double *>(base);
    int32_t *planes = reinterpret_cast<int32_t *>(base + static_cast<size_t>(n_blocks) * sizeof(double));
    int32_t *expo = planes + 6 * static_cast<size_t>(n_blocks);
    const unsigned g = (n_blocks + 3) / 4;
    if (g) hipLaunchKernelGGL(k_seq_blocksum, dim3(g), dim3(256), 0, s, d_chains, n_chains, n_blocks, blk);
    hipLaunchKernelGGL(k_seq_prefix, dim3(n_chains), dim3(64), 0, s, d_chains, n_chains, blk);
    if (g) hipLaunchKernelGGL(k_seq_summary, dim3(g), dim3(256), 0, s, d_chains, n_chains, n_blocks, blk, planes, expo);
    hipLaunchKernelGGL(k_seq_stitch, dim3(n_chains), dim3(64), 0, s, d_chains, n_chains, n_blocks, planes, expo, d_out, d_n_slow);
}

// diagnostics (gbrl_hip_seq_sums): host arrays in, sequential float32 sums out; n_slow = blocks that took the serial fallback
bool seq_sums_selftest(const float *x, const uint32_t *lens, const float *starts, int n_chains, float *out, uint32_t *n_slow_out) {
    if (n_chains <= 0) return true;
    std::vector<SeqChain> ch(n_chains);
    size_t total = 0; uint32_t nb = 0;
    for (int i = 0; i < n_chains; ++i) total += lens[i];
    struct Dev { void *p = nullptr; ~Dev() { if (p) (void)hipFree(p); } };
    Dev dx, dc, ds, dout, dslow;
    if (hipMalloc(&dx.p, std::max<size_t>(16, total * sizeof(float) + 64)) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (total && hipMemcpy(dx.p, x, total * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return false;
    size_t off = 0;
    for (int i = 0; i < n_chains; ++i) {
        ch[i].x = static_cast<const float *>(dx.p) + off; ch[i].len = lens[i]; ch[i].blk0 = nb; ch[i].start = starts ? starts[i] : 0.0f;
        off += lens[i]; nb += (lens[i] + kSeqBlock - 1) / kSeqBlock;
    }
    if (hipMalloc(&dc.p, sizeof(SeqChain) * n_chains) != hipSuccess || hipMalloc(&ds.p, seq_sums_scratch_bytes(std::max(1u, nb))) != hipSuccess ||
        hipMalloc(&dout.p, sizeof(float) * n_chains) != hipSuccess || hipMalloc(&dslow.p, 4) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMemcpy(dc.p, ch.data(), sizeof(SeqChain) * n_chains, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dslow.p, 0, 4) != hipSuccess) return false;
    seq_sums(static_cast<const SeqChain *>(dc.p), n_chains, nb, ds.p, static_cast<float *>(dout.p), static_cast<uint32_t *>(dslow.p), nullptr);
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMemcpy(out, dout.p, sizeof(float) * n_chains, hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (n_slow_out && hipMemcpy(n_slow_out, dslow.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    return true;
}

}  // namespace kern
}  // namespace gbrl
