// small_grow.hip -- RL-sized steps (N <= 8192 rows on one GPU): the WHOLE growth of a tree -- every level's histograms, scores,
// arg-max, row routing, and the leaf sums -- in ONE launch (round 5).
//
// The level-synchronous host loop (engine_step.hip, grow_tree) needs 4-6 dispatches and a host round trip per level; at a few thousand
// rows each of them is launch latency on an idle device, and its dense int64 level buffers [node][F][257][D+1] cost as much for 4096
// rows as for a million.  Here a block owns feature slots (block b: slots b, b + G, ...), keeps the int32 histogram of ONE slot for a
// batch of nodes in its LDS and never writes it out:
//
//   per level, per owned slot, per batch of nodes that fits the LDS:
//     accumulate   hist[node][class][D+1] += (qg[row][0..D) | 1)          LDS atomics, exact integers (wrapping int32 is exact: the
//                                                                         step's fixed-point scale keeps 4096-row sums below 2^31;
//                                                                         4097..8192 rows: int64 accumulators)
//     scan         per (node, 64-class tile): suffix sums over the classes (wave_scan9, DPP), tile totals -> carries, node totals
//     score        one lane per candidate: the SAME fp64 expression as k_score (score_common.h), path / min_data rejections
//     select       greedy: best gain per node; oblivious: per candidate the fp32 sum over the level's nodes IN NODE ORDER, then * w
//   the block's best (per node / per level) goes to global memory, ONE grid barrier, every block reduces all blocks' bests to the
//   winners (same total order as k_argmax_stage1 / k_resolve_splits: higher score, then lower reference index) and routes the rows
//   itself: row -> child from the winner slot's class code.  Every block holds the complete tree state (row -> node map, node tables,
//   paths) in its LDS and reaches the same decisions from the same data, so one barrier per level is the only inter-block exchange.
//   Block 0 mirrors the per-level result block (the layout digest_level reads) into pinned host memory; at the end the leaves are
//   dealt to the blocks, summed (raw gradients, int64 fixed point: k_leaf_sums' arithmetic) and stored to pinned memory, and the last
//   block to finish publishes a sequence word.  The host replays its bookkeeping from the result blocks (as after a device-planned
//   tree) -- one wait per tree.
//
// Everything computed is an exact integer sum or a function of exact integer sums, so the tree is the one the level loop grows, bit
// for bit (tests/test_gpu_small_step.py runs both on every shape; GBRL_HIP_NO_SMALL_GROW=1 is the level loop).
// Reference: fit_greedy_tree (fitter.cpp:263-375), fit_oblivious_tree (:377-484), splitScoreL2 / Cosine (node.cpp:187-251, 321-376),
// splitNode (node.cpp:64-149), calc_leaf_value (fitter.cpp:545-582).
#include "kernels.h"
#include "kernels_common.h"
#include "score_common.h"

#include <algorithm>
#include <cstdlib>

namespace gbrl {
namespace kern {

namespace {

constexpr int kSgThreads = 1024;
constexpr int kSgWaves = kSgThreads / kWave;
constexpr unsigned kLeafBit = 0x8000u;     // rownode: the row has reached a leaf; low 15 bits = node id
constexpr unsigned kRightBit = 0x4000u;    // rownode, between the two routing passes: the row goes right

struct alignas(16) SgBest { float v; int32_t ref; uint32_t slotbin; uint32_t pad; };

__device__ __forceinline__ SgBest sg_better(SgBest a, SgBest b) {
    // (score_common.h `better`, carrying the winner's (slot, bin) along)
    if (b.v > a.v || (b.v == a.v && b.v > -INFINITY && b.ref < a.ref)) return b;
    return a;
}
__device__ __forceinline__ SgBest sg_wave_best(SgBest m) {
    for (int o = kWave / 2; o > 0; o >>= 1) {
        SgBest other;
        other.v = __shfl_xor(m.v, o, kWave);
        other.ref = __shfl_xor(m.ref, o, kWave);
        other.slotbin = static_cast<uint32_t>(__shfl_xor(static_cast<int>(m.slotbin), o, kWave));
        other.pad = 0;
        m = sg_better(m, other);
    }
    return m;
}

__device__ __forceinline__ void sg_add(int32_t *p, int32_t v) { atomicAdd(p, v); }
__device__ __forceinline__ void sg_add(long long *p, long long v) { atomicAdd(reinterpret_cast<unsigned long long *>(p), static_cast<unsigned long long>(v)); }

struct SgLayout {     // byte offsets into the dynamic LDS block (host computes, kernel carves)
    int rownode, scode, tn, tid, psb, pv, nright, split, cidx, cid, win, wcat, nbest, ssum, leafflag, ttot, total, totalf, ibest, wbest, lacc, hist;
    int total_bytes;
};

struct SmallGrowArgs {
    const uint16_t *codes;      // [groups][N][16] class codes
    const int32_t *qg;          // [N][D] fixed-point build gradients
    const float *grads;         // [N][D] raw gradients (leaf sums)
    const StepScales *scales;
    const FeatureSlot *slots;   // [n_slots]
    const float *thr;           // [F][B]
    const float *cand_w;        // [n_cand]
    const int32_t *cand_ref;    // [n_cand]
    int N, D, B, n_slots, NB, MD, min_data, cosine, oblivious;
    int G, NC, nb_cap, Tmax, NIDS;
    uint32_t magicW;            // floor(2^32 / (D + 1)) + 1
    SgBest *bests;              // greedy [MD][NC][G], oblivious [MD][G]
    unsigned *sync;             // [0] barrier arrivals, [1] finished blocks, [2] abort
    char *res;                  // pinned, device-mapped: MD result blocks of res_stride bytes
    int res_stride, max_front;
    int64_t *acc;               // pinned: [NIDS][D+1]
    uint32_t *status;           // pinned: [0] sequence word, [1] levels written, [2] node count, [3] error
    uint32_t seq;
    SgLayout L;
};

// One barrier over the grid (all G blocks are resident: G <= CUs, one block per CU).  Monotonic arrival counter; a timeout (or another
// block's abort) makes every block leave through the error exit instead of spinning forever.
__device__ __forceinline__ bool sg_grid_barrier(unsigned *sync, unsigned G, unsigned &epoch, int *s_abort) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        __threadfence();
        atomicAdd(&sync[0], 1u);
        const unsigned target = epoch * G;
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        int bad = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
                if (__hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bad = 1; break; }
                if (wall_clock64() - t0 > 400000000ll) {   // 4 s of the 100 MHz counter
                    __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bad = 1;
                    break;
                }
            }
        }
        if (bad) *s_abort = 1;
        __threadfence();
    }
    __syncthreads();
    __threadfence();   // every wave: nothing read below may come from a line cached before the barrier
    return *s_abort == 0;
}

template <typename ACC>
__global__ __launch_bounds__(kSgThreads) void k_small_grow(const SmallGrowArgs a) {
    extern __shared__ __align__(16) unsigned char sg_lds[];
    const SgLayout &L = a.L;
    uint16_t *rownode = reinterpret_cast<uint16_t *>(sg_lds + L.rownode);
    uint16_t *scode = reinterpret_cast<uint16_t *>(sg_lds + L.scode);
    int *tn_b = reinterpret_cast<int *>(sg_lds + L.tn);                 // [2][NC] rows of the level's nodes
    int *tid_b = reinterpret_cast<int *>(sg_lds + L.tid);               // [2][NC] their node ids (host numbering)
    uint32_t *psb_b = reinterpret_cast<uint32_t *>(sg_lds + L.psb);     // [2][NC][MD] path: slot << 16 | bin
    float *pv_b = reinterpret_cast<float *>(sg_lds + L.pv);             // [2][NC][MD] path: threshold value
    int *nright = reinterpret_cast<int *>(sg_lds + L.nright);           // [NC]
    int *split = reinterpret_cast<int *>(sg_lds + L.split);             // [NC]
    int *cidx = reinterpret_cast<int *>(sg_lds + L.cidx);               // [NC][2] child's index in the next table, -1: not active
    int *cid = reinterpret_cast<int *>(sg_lds + L.cid);                 // [NC] left child's node id
    SgBest *win = reinterpret_cast<SgBest *>(sg_lds + L.win);           // [NC] the level's winners
    int *wcat = reinterpret_cast<int *>(sg_lds + L.wcat);               // [NC] winner slot is categorical
    SgBest *nbest = reinterpret_cast<SgBest *>(sg_lds + L.nbest);       // [NC] greedy: this block's best per node
    float *ssum = reinterpret_cast<float *>(sg_lds + L.ssum);           // [NB] oblivious: per candidate, sum over nodes
    unsigned char *leafflag = sg_lds + L.leafflag;                      // [NIDS]
    long long *ttot = reinterpret_cast<long long *>(sg_lds + L.ttot);   // [nb][T][W] tile totals, then carries
    long long *total = reinterpret_cast<long long *>(sg_lds + L.total); // [nb][W]
    double *total_f = reinterpret_cast<double *>(sg_lds + L.totalf);    // [nb][W]
    SgBest *ibest = reinterpret_cast<SgBest *>(sg_lds + L.ibest);       // [nb][T]
    SgBest *wbest = reinterpret_cast<SgBest *>(sg_lds + L.wbest);       // [16]
    unsigned long long *lacc = reinterpret_cast<unsigned long long *>(sg_lds + L.lacc);   // [W]
    ACC *hist = reinterpret_cast<ACC *>(sg_lds + L.hist);               // [nb][NBe][W]

    __shared__ int s_abort, s_nact, s_nextid, s_nact_next, s_stop;
    __shared__ SgBest s_bbest;

    const int N = a.N, D = a.D, W = a.D + 1, B = a.B, MD = a.MD, NC = a.NC, G = a.G;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave, blk = blockIdx.x;
    const double inv_scale = a.scales->inv_scale;
    const bool obl = a.oblivious != 0;
    const int NW = N * W, ND = N * D;
    unsigned epoch = 0;

    for (int r = tid; r < N; r += kSgThreads) rownode[r] = 0;
    for (int i = tid; i < a.NIDS; i += kSgThreads) leafflag[i] = 0;
    if (tid == 0) { tn_b[0] = N; tid_b[0] = 0; s_abort = 0; s_nact = 1; s_nextid = 1; s_stop = 0; }
    __syncthreads();

    int cur = 0, level = 0, loaded_slot = -1;
    bool ok = true;
    for (; level < MD; ++level) {
        const int n_act = s_nact;
        int *tn = tn_b + cur * NC, *tidc = tid_b + cur * NC;
        uint32_t *psb = psb_b + static_cast<size_t>(cur) * NC * MD;
        float *pv = pv_b + static_cast<size_t>(cur) * NC * MD;
        // ---- this block's slots ------------------------------------------------------------------------------------------------
        if (obl) { if (tid == 0) s_bbest = SgBest{-INFINITY, 0x7fffffff, 0u, 0u}; }
        else for (int k = tid; k < n_act; k += kSgThreads) nbest[k] = SgBest{-INFINITY, 0x7fffffff, 0u, 0u};
        for (int fs = blk; fs < a.n_slots; fs += G) {
            const FeatureSlot sl = a.slots[fs];
            const int NBe = sl.n_cand + 1;                       // classes of this slot
            const int T = (NBe - 1 + kWave - 1) / kWave;         // 64-class tiles over the classes 1 .. NBe-1
            if (fs != loaded_slot) {                             // the slot's class codes in row order
                const uint16_t *cs = a.codes + (static_cast<size_t>(fs >> 4) * N) * kCodeGroup + (fs & (kCodeGroup - 1));
                for (int r = tid; r < N; r += kSgThreads) scode[r] = cs[static_cast<size_t>(r) * kCodeGroup];
                loaded_slot = fs;
            }
            if (obl) for (int k = tid; k < sl.n_cand; k += kSgThreads) ssum[k] = 0.0f;
            for (int k0 = 0; k0 < n_act; k0 += a.nb_cap) {
                const int nbk = min(a.nb_cap, n_act - k0);
                const int hwords = nbk * NBe * W;
                for (int i = tid; i < hwords; i += kSgThreads) hist[i] = 0;
                __syncthreads();
                // -- accumulate: element e = (row, field); loads first, atomics second
                constexpr int U = 4;
                for (int e0 = tid; e0 < NW; e0 += kSgThreads * U) {
                    int rr[U], jj[U], qq[U];
                    unsigned kk[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int e = e0 + u * kSgThreads;
                        const int ee = min(e, NW - 1);
                        rr[u] = static_cast<int>(__umulhi(static_cast<unsigned>(ee), a.magicW));
                        jj[u] = ee - rr[u] * W;
                        qq[u] = a.qg[min(ee - rr[u], ND - 1)];
                        kk[u] = e < NW ? rownode[rr[u]] : 0xffffu;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const unsigned kb = kk[u] - static_cast<unsigned>(k0);
                        if (kb < static_cast<unsigned>(nbk)) {
                            const int c = scode[rr[u]];
                            const int v = jj[u] < D ? qq[u] : 1;
                            sg_add(&hist[(kb * NBe + c) * W + jj[u]], static_cast<ACC>(v));
                        }
                    }
                }
                __syncthreads();
                // -- phase A: per (node, tile) suffix sums over the tile's classes (in place) and the tile totals
                const int n_items = nbk * T;
                for (int it = wave; it < n_items; it += kSgWaves) {
                    const int kb = it / T, t = it - kb * T;
                    const int c = NBe - 1 - (t * kWave + lane);   // this lane's class; the tile's lanes run DOWN the classes
                    const bool have = c >= 1;
                    ACC *hc = hist + (static_cast<size_t>(kb) * NBe + (have ? c : 0)) * W;
                    for (int w0 = 0; w0 < W; w0 += 9) {
                        long long v[9];
#pragma unroll
                        for (int j = 0; j < 9; ++j) v[j] = (have && w0 + j < W) ? static_cast<long long>(hc[min(w0 + j, W - 1)]) : 0;
                        wave_scan9(v);
                        if (!sl.is_cat) {
#pragma unroll
                            for (int j = 0; j < 9; ++j) if (have && w0 + j < W) hc[w0 + j] = static_cast<ACC>(v[j]);
                        }
                        if (lane == kWave - 1) {
#pragma unroll
                            for (int j = 0; j < 9; ++j) if (w0 + j < W) ttot[(static_cast<size_t>(kb) * T + t) * W + w0 + j] = v[j];
                        }
                    }
                }
                __syncthreads();
                // -- carries (sum of the tiles above) and node totals (all tiles + class 0)
                for (int i = tid; i < nbk * W; i += kSgThreads) {
                    const int kb = i / W, j = i - kb * W;
                    long long run = 0;
                    for (int t = 0; t < T; ++t) {
                        long long *p = &ttot[(static_cast<size_t>(kb) * T + t) * W + j];
                        const long long x = *p;
                        *p = run;
                        run += x;
                    }
                    run += static_cast<long long>(hist[(static_cast<size_t>(kb) * NBe) * W + j]);
                    total[i] = run;
                    total_f[i] = static_cast<double>(run);
                }
                __syncthreads();
                // -- phase B: one lane per candidate
                for (int it = wave; it < n_items; it += kSgWaves) {
                    const int kb = it / T, t = it - kb * T;
                    const int k_abs = k0 + kb;
                    const int c = NBe - 1 - (t * kWave + lane);
                    const int k = c - 1;                          // candidate: numeric threshold k (right = classes > k), categorical class k + 1
                    const bool have = c >= 1;
                    const long long *tot = total + static_cast<size_t>(kb) * W;
                    const double *totf = total_f + static_cast<size_t>(kb) * W;
                    const long long n_tot = tot[D];
                    // the node's path conditions on this slot (a candidate that repeats one is rejected, node.cpp:154-166)
                    int np = 0;
                    for (int p = 0; p < level; ++p) np += (psb[k_abs * MD + p] >> 16) == static_cast<uint32_t>(fs) ? 1 : 0;
                    ACC *hc = hist + (static_cast<size_t>(kb) * NBe + (have ? c : 0)) * W;
                    const long long *cy = ttot + (static_cast<size_t>(kb) * T + t) * W;
                    const bool scanned = !sl.is_cat;
                    float out = -INFINITY;
                    if (have) {
                        const long long n_r = static_cast<long long>(hc[D]) + (scanned ? cy[D] : 0), n_l = n_tot - n_r;
                        bool reject = (n_l < a.min_data) || (n_r < a.min_data);
                        if (np > 0) {
                            const float tk = sl.is_cat ? 0.0f : a.thr[static_cast<size_t>(fs) * B + k];
                            for (int p = 0; p < level; ++p) {
                                const uint32_t sb = psb[k_abs * MD + p];
                                if ((sb >> 16) != static_cast<uint32_t>(fs)) continue;
                                if (sl.is_cat) reject |= static_cast<int>(sb & 0xffffu) == k + 1;
                                else reject |= pv[k_abs * MD + p] == tk;
                            }
                        }
                        if (!reject)
                            out = candidate_score([&](int d) { return static_cast<double>(static_cast<long long>(hc[d]) + (scanned ? cy[d] : 0)); }, totf, D, n_l, n_r,
                                                  a.cosine, inv_scale);
                    }
                    if (obl) {
                        if (have) hc[0] = static_cast<ACC>(__float_as_int(out));   // the score takes the place of a word nobody reads again
                    } else {
                        // greedy: gain = fma(score, w, -parent), root parent = 0 (fitter.cpp:315-316, 332); parent score from the node totals
                        float par_sub = 0.0f;
                        if (level > 0) {
                            const double x = side_term(tot, D, n_tot, inv_scale);
                            par_sub = static_cast<float>(a.cosine ? sqrt(x) : x);
                        }
                        SgBest mine{-INFINITY, 0x7fffffff, 0u, 0u};
                        if (have) {
                            const int j = sl.cand_base + k;
                            mine = SgBest{fmaf(out, a.cand_w[j], -par_sub), a.cand_ref[j],
                                          (static_cast<uint32_t>(fs) << 16) | static_cast<uint32_t>(sl.is_cat ? k + 1 : k), 0u};
                        }
                        mine = sg_wave_best(sg_better(SgBest{-INFINITY, 0x7fffffff, 0u, 0u}, mine));
                        if (lane == 0) ibest[it] = mine;
                    }
                }
                __syncthreads();
                // -- phase C
                if (obl) {
                    for (int k = tid; k < sl.n_cand; k += kSgThreads) {   // fp32 sum over the level's nodes in node order (fitter.cpp:426-435)
                        float s = ssum[k];
                        for (int kb = 0; kb < nbk; ++kb) s += __int_as_float(static_cast<int>(hist[(static_cast<size_t>(kb) * NBe + k + 1) * W]));
                        ssum[k] = s;
                    }
                } else {
                    for (int kb = tid; kb < nbk; kb += kSgThreads) {
                        SgBest b = nbest[k0 + kb];
                        for (int t = 0; t < T; ++t) b = sg_better(b, ibest[kb * T + t]);
                        nbest[k0 + kb] = b;
                    }
                }
                __syncthreads();
            }
            if (obl) {   // the slot's best candidate: (sum over nodes) * w, lowest reference index among maxima (fitter.cpp:435-444)
                SgBest mine{-INFINITY, 0x7fffffff, 0u, 0u};
                for (int k = tid; k < sl.n_cand; k += kSgThreads) {
                    const int j = sl.cand_base + k;
                    mine = sg_better(mine, SgBest{ssum[k] * a.cand_w[j], a.cand_ref[j], (static_cast<uint32_t>(fs) << 16) | static_cast<uint32_t>(sl.is_cat ? k + 1 : k), 0u});
                }
                mine = sg_wave_best(mine);
                if (lane == 0) wbest[wave] = mine;
                __syncthreads();
                if (tid == 0) {
                    SgBest b = s_bbest;
                    for (int q = 0; q < kSgWaves; ++q) b = sg_better(b, wbest[q]);
                    s_bbest = b;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        // ---- publish this block's bests, ONE grid barrier, reduce to the winners -------------------------------------------------
        SgBest *lv = a.bests + static_cast<size_t>(level) * (obl ? 1 : NC) * G;
        if (obl) { if (tid == 0) lv[blk] = s_bbest; }
        else for (int k = tid; k < n_act; k += kSgThreads) lv[static_cast<size_t>(k) * G + blk] = nbest[k];
        if (!sg_grid_barrier(a.sync, static_cast<unsigned>(G), epoch, &s_abort)) { ok = false; break; }
        const int n_win = obl ? 1 : n_act;
        for (int k = wave; k < n_win; k += kSgWaves) {
            SgBest m{-INFINITY, 0x7fffffff, 0u, 0u};
            for (int q = lane; q < G; q += kWave) {
                typedef unsigned int sg_u4 __attribute__((ext_vector_type(4)));
                const sg_u4 raw = __builtin_nontemporal_load(reinterpret_cast<const sg_u4 *>(lv + static_cast<size_t>(k) * G + q));
                m = sg_better(m, SgBest{__uint_as_float(raw.x), static_cast<int32_t>(raw.y), raw.z, 0u});
            }
            m = sg_wave_best(m);
            if (lane == 0) {
                win[k] = m;
                wcat[k] = m.ref == 0x7fffffff ? 0 : a.slots[m.slotbin >> 16].is_cat;
            }
        }
        for (int k = tid; k < n_act; k += kSgThreads) nright[k] = 0;
        __syncthreads();
        for (int k = tid; k < n_act; k += kSgThreads) {
            const SgBest w = win[obl ? 0 : k];
            // fitter.cpp:458 oblivious: grow while any candidate is finite; fitter.cpp:357 greedy: split iff best >= 0
            split[k] = obl ? (w.v != -INFINITY ? 1 : 0) : (w.v >= 0.0f ? 1 : 0);
        }
        __syncthreads();
        // ---- routing pass 1: which side every row of a splitting node takes; rows going right per node ---------------------------
        const int rounds = (N + kSgThreads - 1) / kSgThreads;
        for (int q = 0; q < rounds; ++q) {
            const int r = q * kSgThreads + tid;
            const unsigned v = r < N ? rownode[r] : 0xffffu;
            const bool act = v < kLeafBit;
            const int k = act ? static_cast<int>(v) : 0;
            const bool sp = act && split[k] != 0;
            bool right = false;
            if (sp) {
                const SgBest w = win[obl ? 0 : k];
                const int slot = static_cast<int>(w.slotbin >> 16), bin = static_cast<int>(w.slotbin & 0xffffu);
                const int code = a.codes[(static_cast<size_t>(slot >> 4) * N + r) * kCodeGroup + (slot & (kCodeGroup - 1))];
                right = wcat[obl ? 0 : k] ? (code == bin) : (code > bin);
                if (right) rownode[r] = static_cast<uint16_t>(v | kRightBit);
            }
            // (a wave whose rows all sit in one node -- the usual case near the root -- counts with one atomic)
            const int k_first = __builtin_amdgcn_readfirstlane(act ? k : -1);
            const unsigned long long m_right = __ballot(right);
            if (__all(!right || k == k_first)) {
                if (lane == 0 && m_right) atomicAdd(&nright[k_first], __popcll(m_right));
            } else if (right) {
                atomicAdd(&nright[k], 1);
            }
        }
        __syncthreads();
        // ---- children: node ids in the host's order (for k in splitting: left, right), next level's table ------------------------
        const bool last_level = level + 1 == MD;
        int *tn2 = tn_b + (cur ^ 1) * NC, *tid2 = tid_b + (cur ^ 1) * NC;
        uint32_t *psb2 = psb_b + static_cast<size_t>(cur ^ 1) * NC * MD;
        float *pv2 = pv_b + static_cast<size_t>(cur ^ 1) * NC * MD;
        if (wave == 0) {
            int run_split = 0, run_act = 0;
            const int next_id = s_nextid;
            const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (kWave - lane));
            for (int base = 0; base < n_act; base += kWave) {
                const int k = base + lane;
                const bool valid = k < n_act;
                const bool sp = valid && split[k] != 0;
                const unsigned long long m = __ballot(sp);
                const int my_id = next_id + 2 * (run_split + __popcll(m & lt));
                const int nr = valid ? nright[k] : 0, nl = valid ? tn[k] - nr : 0;
                const bool al = sp && !last_level && (obl || nl > 0), ar = sp && !last_level && (obl || nr > 0);
                const unsigned long long ml = __ballot(al), mr = __ballot(ar);
                const int il = run_act + __popcll(ml & lt) + __popcll(mr & lt), ir = il + (al ? 1 : 0);
                if (valid) {
                    cid[k] = my_id;
                    cidx[2 * k] = al ? il : -1;
                    cidx[2 * k + 1] = ar ? ir : -1;
                    if (al) { tn2[il] = nl; tid2[il] = my_id; }
                    if (ar) { tn2[ir] = nr; tid2[ir] = my_id + 1; }
                    if (!sp) leafflag[tidc[k]] = 1;                            // the node stays a leaf
                    else {
                        if (!al) leafflag[my_id] = 1;                          // a child that is not grown further is a leaf (empty, or the last level)
                        if (!ar) leafflag[my_id + 1] = 1;
                    }
                }
                run_split += __popcll(m);
                run_act += __popcll(ml) + __popcll(mr);
            }
            if (lane == 0) { s_nextid = next_id + 2 * run_split; s_nact_next = run_act; s_stop = run_split == 0 ? 1 : 0; }
        }
        __syncthreads();
        // paths of the active children: the parent's, plus the new condition
        for (int i = tid; i < n_act * 2 * (level + 1); i += kSgThreads) {
            const int p = i % (level + 1), ks = i / (level + 1), k = ks >> 1, side = ks & 1;
            const int dst = split[k] ? cidx[2 * k + side] : -1;
            if (dst < 0) continue;
            if (p < level) { psb2[dst * MD + p] = psb[k * MD + p]; pv2[dst * MD + p] = pv[k * MD + p]; }
            else {
                const SgBest w = win[obl ? 0 : k];
                const int slot = static_cast<int>(w.slotbin >> 16), bin = static_cast<int>(w.slotbin & 0xffffu);
                psb2[dst * MD + p] = w.slotbin;
                pv2[dst * MD + p] = wcat[obl ? 0 : k] ? INFINITY : a.thr[static_cast<size_t>(slot) * B + bin];
            }
        }
        // block 0: the level's result block for the host (the layout digest_level reads) + the winners' threshold values
        if (blk == 0) {
            char *res = a.res + static_cast<size_t>(level) * a.res_stride;
            int32_t *r_idx = reinterpret_cast<int32_t *>(res);
            float *r_score = reinterpret_cast<float *>(res + 4 * static_cast<size_t>(a.max_front));
            int64_t *r_cnt = reinterpret_cast<int64_t *>(res + 8 * static_cast<size_t>(a.max_front));
            float *r_thr = reinterpret_cast<float *>(res + 40 * static_cast<size_t>(a.max_front));
            for (int k = tid; k < n_act; k += kSgThreads) {
                const SgBest w = win[obl ? 0 : k];
                if (!obl || k == 0) { r_idx[k] = w.ref == 0x7fffffff ? 0 : w.ref; r_score[k] = w.v; }
                r_cnt[k] = tn[k];
                r_cnt[static_cast<size_t>(a.max_front) + k] = nright[k];
                float tv = 0.0f;
                if (w.ref != 0x7fffffff && !wcat[obl ? 0 : k]) tv = a.thr[static_cast<size_t>(w.slotbin >> 16) * B + (w.slotbin & 0xffffu)];
                r_thr[k] = tv;
            }
        }
        // ---- routing pass 2: rows move to their child (or stay with a node that has become a leaf) -------------------------------
        for (int r = tid; r < N; r += kSgThreads) {
            const unsigned v = rownode[r];
            if (v & kLeafBit) continue;
            const int k = static_cast<int>(v & 0x3fffu), side = (v & kRightBit) ? 1 : 0;
            unsigned nv;
            if (!split[k]) nv = kLeafBit | static_cast<unsigned>(tidc[k]);
            else {
                const int ci = cidx[2 * k + side];
                nv = ci >= 0 ? static_cast<unsigned>(ci) : (kLeafBit | static_cast<unsigned>(cid[k] + side));
            }
            rownode[r] = static_cast<uint16_t>(nv);
        }
        __syncthreads();
        const int stop = s_stop;
        if (tid == 0) s_nact = s_nact_next;
        cur ^= 1;
        __syncthreads();
        if (stop || last_level) { ++level; break; }
    }
    // Oblivious trees keep growing with EMPTY nodes in the table; whatever is still active when the loop ends is a leaf.
    if (ok) {
        const int n_act = s_nact;
        const int *tidc = tid_b + cur * NC;
        for (int k = tid; k < n_act; k += kSgThreads) leafflag[tidc[k]] = 1;
        for (int r = tid; r < N; r += kSgThreads) {
            const unsigned v = rownode[r];
            if (!(v & kLeafBit)) rownode[r] = static_cast<uint16_t>(kLeafBit | static_cast<unsigned>(tidc[v & 0x3fffu]));
        }
        __syncthreads();
        // ---- leaf sums of the RAW gradients (k_leaf_sums' arithmetic: int64 fixed point, exact), leaves dealt to the blocks -----------
        const double leaf_scale = a.scales->leaf_scale;
        const int n_ids = s_nextid;
        const int per = kSgThreads / D > 0 ? kSgThreads / D : 1;
        const int d = tid % D, sub = tid / D;
        for (int id = blk; id < n_ids; id += G) {
            if (!leafflag[id]) continue;
            if (tid < W) lacc[tid] = 0ull;
            __syncthreads();
            if (sub < per) {
                long long s = 0;
                int cnt = 0;
                const unsigned want = kLeafBit | static_cast<unsigned>(id);
                for (int r = sub; r < N; r += per) {
                    if (rownode[r] == want) {
                        s += __double2ll_rn(static_cast<double>(a.grads[static_cast<size_t>(r) * D + d]) * leaf_scale);
                        ++cnt;
                    }
                }
                if (cnt) {
                    atomicAdd(&lacc[d], static_cast<unsigned long long>(s));
                    if (d == 0) atomicAdd(&lacc[D], static_cast<unsigned long long>(cnt));
                }
            }
            __syncthreads();
            if (tid < W) a.acc[static_cast<size_t>(id) * W + tid] = static_cast<int64_t>(lacc[tid]);
            __syncthreads();
        }
    }
    // ---- the last block to finish publishes the sequence word (and hands the counters back zeroed) --------------------------------
    __syncthreads();
    if (tid == 0) {
        if (blk == 0) { a.status[1] = static_cast<uint32_t>(level); a.status[2] = static_cast<uint32_t>(s_nextid); }
        __threadfence_system();
        if (atomicAdd(&a.sync[1], 1u) == static_cast<unsigned>(G) - 1u) {
            const unsigned aborted = __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.sync[0] = 0u; a.sync[1] = 0u; a.sync[2] = 0u;
            a.status[3] = aborted || !ok ? 1u : 0u;
            __threadfence_system();
            __hip_atomic_store(a.status, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int align16(int x) { return (x + 15) & ~15; }

}  // namespace

// ---------------------------------------------------------------------------------------------------- host side

size_t small_grow_bests_bytes(int MD, int G, bool oblivious) {
    const int NC = 1 << std::max(0, MD - 1);
    return sizeof(SgBest) * static_cast<size_t>(MD) * (oblivious ? 1 : NC) * G;
}
size_t small_grow_res_stride(int MD) { return static_cast<size_t>(1 << std::max(0, MD - 1)) * 44 + 64; }

int small_grow_blocks(int n_slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    const int cap = [] { const char *e = std::getenv("GBRL_HIP_SMALL_GROW_BLOCKS"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 1 << 30; }();   /* read per call: the tests flip it */   // test / measurement hook
    return std::max(1, std::min(std::min(n_slots, cus), cap));
}

namespace {
// LDS layout for one launch; nb_cap = 0: the shape does not fit
SgLayout sg_layout(int N, int D, int NB, int MD, int acc_bytes, int &nb_cap, int &Tmax) {
    const int W = D + 1, NC = 1 << std::max(0, MD - 1), NIDS = 2 << MD;
    Tmax = std::max(1, (NB - 1 + kWave - 1) / kWave);
    SgLayout L{};
    int off = 0;
    auto take = [&](int bytes) { const int o = off; off = align16(off + bytes); return o; };
    L.rownode = take(2 * N);
    L.scode = take(2 * N);
    L.tn = take(4 * 2 * NC);
    L.tid = take(4 * 2 * NC);
    L.psb = take(4 * 2 * NC * MD);
    L.pv = take(4 * 2 * NC * MD);
    L.nright = take(4 * NC);
    L.split = take(4 * NC);
    L.cidx = take(8 * NC);
    L.cid = take(4 * NC);
    L.win = take(16 * NC);
    L.wcat = take(4 * NC);
    L.nbest = take(16 * NC);
    L.ssum = take(4 * NB);
    L.leafflag = take(NIDS);
    L.wbest = take(16 * kSgWaves);
    L.lacc = take(8 * W);
    const int budget = 160 * 1024 - 1024 /* static __shared__ + slack */ - off;
    const int per_node = NB * W * acc_bytes + Tmax * W * 8 + W * 16 + Tmax * 16 + 64;
    nb_cap = budget > 0 ? std::min(NC, budget / per_node) : 0;
    if (nb_cap <= 0) { nb_cap = 0; return L; }
    L.ttot = take(8 * nb_cap * Tmax * W);
    L.total = take(8 * nb_cap * W);
    L.totalf = take(8 * nb_cap * W);
    L.ibest = take(16 * nb_cap * Tmax);
    L.hist = take(acc_bytes * nb_cap * NB * W);
    L.total_bytes = off;
    return L;
}
}  // namespace

bool small_grow_supported(int N, int D, int NB, int MD, int n_slots, int n_cand) {
    if (N < 1 || N > 8192 || D < 1 || D > 512 || MD < 1 || MD > 8 || n_slots < 1 || n_slots > 65535 || NB < 2 || NB > 65535 || n_cand < 1) return false;
    int nb = 0, T = 0;
    (void)sg_layout(N, D, NB, MD, N <= 4096 ? 4 : 8, nb, T);
    return nb >= 1;
}

bool small_grow(const SmallGrowIO &io, hipStream_t s) {
    const int acc_bytes = io.N <= 4096 ? 4 : 8;
    SmallGrowArgs a{};
    a.L = sg_layout(io.N, io.D, io.NB, io.MD, acc_bytes, a.nb_cap, a.Tmax);
    if (a.nb_cap < 1) return false;
    a.codes = io.codes; a.qg = io.qg; a.grads = io.grads; a.scales = io.scales; a.slots = io.slots; a.thr = io.thr; a.cand_w = io.cand_w; a.cand_ref = io.cand_ref;
    a.N = io.N; a.D = io.D; a.B = io.B; a.n_slots = io.n_slots; a.NB = io.NB; a.MD = io.MD; a.min_data = io.min_data; a.cosine = io.cosine ? 1 : 0; a.oblivious = io.oblivious ? 1 : 0;
    a.G = io.G; a.NC = 1 << std::max(0, io.MD - 1); a.NIDS = 2 << io.MD;
    a.magicW = static_cast<uint32_t>((1ull << 32) / static_cast<unsigned>(io.D + 1)) + 1u;
    a.bests = static_cast<SgBest *>(io.bests); a.sync = io.sync; a.res = io.res; a.res_stride = static_cast<int>(small_grow_res_stride(io.MD)); a.max_front = a.NC;
    a.acc = io.acc; a.status = io.status; a.seq = io.seq;
    static PerDeviceOnce attr32, attr64;
    if (acc_bytes == 4) {
        if (attr32.first() && hipFuncSetAttribute(reinterpret_cast<const void *>(k_small_grow<int32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) { (void)hipGetLastError(); attr32.done = 0; return false; }
        hipLaunchKernelGGL(k_small_grow<int32_t>, dim3(io.G), dim3(kSgThreads), a.L.total_bytes, s, a);
    } else {
        if (attr64.first() && hipFuncSetAttribute(reinterpret_cast<const void *>(k_small_grow<long long>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) { (void)hipGetLastError(); attr64.done = 0; return false; }
        hipLaunchKernelGGL(k_small_grow<long long>, dim3(io.G), dim3(kSgThreads), a.L.total_bytes, s, a);
    }
    return hipGetLastError() == hipSuccess;
}

}  // namespace kern
}  // namespace gbrl
