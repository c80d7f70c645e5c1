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
#include "hooks.h"
#include "kernels_common.h"
#include "score_common.h"
#include "neartie_core.h"

#include <algorithm>
#include <cstdlib>

namespace gbrl {
namespace kern {

namespace {

constexpr int kSgThreads = 1024;
constexpr int kSgWaves = kSgThreads / kWave;
constexpr unsigned kLeafBit = 0x8000u;     // rownode: the row has reached a leaf; low 15 bits = node id

// One candidate in the arg-max: gain / score, reference candidate index (ties: the lower wins), feature slot << 16 | bin (numeric: threshold
// index; categorical: class), and -- greedy growth -- `pad` = the rows the candidate sends RIGHT in its node (known where the candidate is
// scored; travels with the winner so that nobody has to count the children's rows; at most 8192, bit 31 carries the near-tie flag between blocks).
struct alignas(16) SgBest { float v; int32_t ref; uint32_t slotbin; uint32_t pad; };

__device__ __forceinline__ SgBest sg_better(SgBest a, SgBest b) {
    // (score_common.h `better`, carrying the winner's (slot, bin) along)
    if (b.v > a.v || (b.v == a.v && b.v > -INFINITY && b.ref < a.ref)) return b;
    return a;
}
// Reduction over the 64 lanes with DPP moves (row_shr 1, 2, 4, 8, then row_bcast 15 / 31: lane 63 ends up with the best of all) and three
// readlanes -- no LDS crossbar round trips (a __shfl_xor butterfly is 18 ds_bpermute with ~100 clocks of latency each, and these
// reductions sit on the critical path of every level).  A lane without a source keeps its own value (update_dpp's `old`).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ SgBest sg_dpp_step(SgBest m) {
    SgBest o;
    o.v = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m.v), __float_as_int(m.v), CTRL, ROW_MASK, 0xf, false));
    o.ref = __builtin_amdgcn_update_dpp(m.ref, m.ref, CTRL, ROW_MASK, 0xf, false);
    o.slotbin = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(m.slotbin), static_cast<int>(m.slotbin), CTRL, ROW_MASK, 0xf, false));
    o.pad = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(m.pad), static_cast<int>(m.pad), CTRL, ROW_MASK, 0xf, false));
    return sg_better(m, o);
}
__device__ __forceinline__ SgBest sg_wave_best(SgBest m) {
    m = sg_dpp_step<0x111, 0xf>(m);   // row_shr:1
    m = sg_dpp_step<0x112, 0xf>(m);   // row_shr:2
    m = sg_dpp_step<0x114, 0xf>(m);   // row_shr:4
    m = sg_dpp_step<0x118, 0xf>(m);   // row_shr:8
    m = sg_dpp_step<0x142, 0xa>(m);   // row_bcast:15 -> rows 1, 3
    m = sg_dpp_step<0x143, 0xc>(m);   // row_bcast:31 -> rows 2, 3
    SgBest r;
    r.v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m.v), 63));
    r.ref = __builtin_amdgcn_readlane(m.ref, 63);
    r.slotbin = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m.slotbin), 63));
    r.pad = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m.pad), 63));
    return r;
}

// Near-tie detection: the wave's best value STRICTLY below `top` (the wave's maximum, already known to every lane) over the lanes' (v, sec)
// pairs -- sec = a lane's own best below its v.  A plain maximum with the same DPP steps (one v_max_f32 with a DPP operand per step).  The
// kernel takes batches of <= 8192 rows, where candidates are told apart by their gain alone (score_common.h near_class).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float sg_dpp_max_step(float x) {
    return fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROW_MASK, 0xf, false)));
}
__device__ __forceinline__ float sg_wave_second(float v, float sec, float top) {
    float x = v < top ? fmaxf(v, sec) : sec;
    x = sg_dpp_max_step<0x111, 0xf>(x);
    x = sg_dpp_max_step<0x112, 0xf>(x);
    x = sg_dpp_max_step<0x114, 0xf>(x);
    x = sg_dpp_max_step<0x118, 0xf>(x);
    x = sg_dpp_max_step<0x142, 0xa>(x);
    x = sg_dpp_max_step<0x143, 0xc>(x);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

__device__ __forceinline__ void sg_add(int32_t *p, int32_t v) { atomicAdd(p, v); }
__device__ __forceinline__ void sg_add(long long *p, long long v) { atomicAdd(reinterpret_cast<unsigned long long *>(p), static_cast<unsigned long long>(v)); }

// hist[(node - k0) * NBe + class][0 .. D] += (qg[row][0 .. D) | 1) for the rows whose node is in the batch [k0, k0 + nbk).
// V = 4 / 2 / 1 gradient words per load (D % 4 == 0, D % 2 == 0, any D): a lane reads its row's fields with 16- / 8- / 4-byte loads.
template <typename ACC, int V>
__device__ __forceinline__ void sg_accumulate_rows(const int32_t *__restrict__ qg, const uint32_t *rc, ACC *hist, int N, int D, int NBe, int k0, int nbk, int tid) {
    typedef int sg_iv __attribute__((ext_vector_type(V)));
    const int W = D + 1;
    constexpr int R = 4;
    for (int r0 = tid; r0 < N; r0 += kSgThreads * R) {
        ACC *base[R];
        const int32_t *src[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int r = r0 + u * kSgThreads, rr = min(r, N - 1);
            const uint32_t rv = rc[rr];
            const unsigned kb = (rv & 0xffffu) - static_cast<unsigned>(k0);
            const bool okk = r < N && kb < static_cast<unsigned>(nbk);
            base[u] = okk ? hist + (static_cast<size_t>(kb) * NBe + (rv >> 16)) * W : nullptr;
            src[u] = qg + static_cast<size_t>(rr) * D;
        }
        for (int j0 = 0; j0 < D; j0 += V) {
            sg_iv q[R];
#pragma unroll
            for (int u = 0; u < R; ++u) q[u] = *reinterpret_cast<const sg_iv *>(src[u] + j0);
#pragma unroll
            for (int u = 0; u < R; ++u) {
                if (base[u]) {
#pragma unroll
                    for (int i = 0; i < V; ++i) sg_add(base[u] + j0 + i, static_cast<ACC>(q[u][i]));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) if (base[u]) sg_add(base[u] + D, static_cast<ACC>(1));
    }
}

// In-kernel near-tie replay (greedy trees, one flagged node per level): a candidate inside the window, as the blocks exchange it
struct alignas(16) SgCand { float gain; int32_t ref; uint32_t slotbin; uint32_t nr; };
constexpr int kSgCandCap = 32;        // per block (distinct neighbours inside the window)
constexpr int kSgMergeCap = 64;       // over all blocks
struct SgNearScratch {                // device memory, small_grow_near_bytes(G, N)
    uint32_t *count;                  // [G] candidates every block collected (0xffff: too many)
    SgCand *list;                     // [G][kSgCandCap]
    float *rep;                       // [kNearCands + 1] replayed scores, [kNearCands] = the parent's
    int32_t *ent;                     // [G][N] ordered row lists, one region per block
};

struct SgLayout {     // byte offsets into the dynamic LDS block (host computes, kernel carves)
    int rc, sw, sref, tn, tid, psb, pv, split, cidx, cid, win, wcat, wthr, nbest, nbest2, npar, ssum, leafflag, ttot, total, totalf, ibest, ibest2, wbest, wbest2, lacc, hist;
    int total_bytes;
};

struct SmallGrowArgs {
    const uint16_t *codes;      // [groups][N][16] class codes
    const uint16_t *codes_fm;   // nullable: [n_fm][N] feature-major copy of the codes of the slots < n_fm (coalesced column loads and row routing)
    int n_fm;
    const int32_t *qg;          // [N][D] fixed-point build gradients
    const float *grads;         // [N][D] raw gradients (leaf sums)
    const StepScales *scales;
    const FeatureSlot *slots;   // [n_slots]
    const float *thr;           // [F][B]
    int n_thr_slots;            // F: the slots that have thresholds
    const float *cand_w;        // [n_cand]
    const int32_t *cand_ref;    // [n_cand]
    int N, D, B, n_slots, NB, MD, min_data, cosine, oblivious;
    int G, NC, nb_cap, Tmax, NIDS;
    uint32_t magicW;            // floor(2^32 / (D + 1)) + 1
    SgBest *bests;              // greedy [MD][NC][G], oblivious [MD][G]
    float near_rel;             // 0: no detection
    SgNearScratch nt;           // in-kernel replay
    int near_in_kernel;         // the idle histogram region holds the replay's scratch (else a flagged tree goes to the level loop)
    int near_tile;              // floats of that scratch the replay stages a batch of rows in
    uint32_t *ckpt;             // device: the tree state the default variant leaves when it gives a flagged tree up -- [16] header, then the fixed part of its LDS
    int ckpt_words;             // 32-bit words of that fixed part (SgLayout::ttot / 4)
    int resume;                 // REPLAY variant: continue from the checkpoint (the flagged level's second pass) instead of growing from the root
    const float *meanden;       // L2: the step's standardisation (mean | std + 1e-8f), nullptr for Cosine
    int tiny_words;             // 32-bit words of the histogram region every wave may use as scratch (sg_tiny_zero_gain)
    unsigned *sync;             // [0] groups arrived, [1] finished blocks, [2] abort, [32 + 32 g] arrivals of group g (kSmallGrowSyncBytes)
    char *res;                  // pinned, device-mapped: MD result blocks of res_stride bytes (block 0 copies them out of res_dev at the end)
    char *res_dev;              // device staging of the same
    int res_stride, max_front;
    int64_t *acc;               // pinned: [NIDS][D+1]
    uint32_t *status;           // pinned: [0] sequence word, [1] levels written, [2] node count, [3] 1 = error, 2 = a near-tie was met (the level loop grows this tree), [4] levels replayed in here
    uint32_t seq;
    StepScales *scales_out;     // pinned (nullable): block 0 mirrors the step's scales for the host
    uint32_t *prof;             // nullable (GBRL_HIP_SMALL_GROW_PROF=1): block 0's time per phase, 10 ns units, 16 words, pinned
    SgLayout L;
};

// One barrier over the grid (all G blocks are resident: G <= CUs, one block per CU).  Monotonic arrival counter; a timeout (or another
// block's abort) makes every block leave through the error exit instead of spinning forever.
// NO fences: an agent-scope release / acquire fence is a write-back / invalidate of the XCD's whole L2 (buffer_wbl2 / buffer_inv), and with
// 16 waves of 192 blocks executing one per level the first version spent 40 us per level behind them (and every block re-fetched its
// gradients from HBM afterwards).  The only data that crosses blocks are the per-level bests: they are written and read with agent-scope
// atomic accesses (sc1: written through to / fetched from the coherence point), the writers drain their stores (s_waitcnt vmcnt(0)) in
// front of the block barrier that precedes the arrival, and the arrival itself is an agent-scope atomic.
__device__ __forceinline__ void sg_store_best(SgBest *p, SgBest b) {
    unsigned long long lo = (static_cast<unsigned long long>(static_cast<uint32_t>(b.ref)) << 32) | __float_as_uint(b.v);
    unsigned long long hi = (static_cast<unsigned long long>(b.pad) << 32) | b.slotbin;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ SgBest sg_load_best(const SgBest *p) {
    const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return SgBest{__uint_as_float(static_cast<uint32_t>(lo)), static_cast<int32_t>(static_cast<uint32_t>(lo >> 32)), static_cast<uint32_t>(hi), static_cast<uint32_t>(hi >> 32)};
}
// Arrivals are counted per GROUP of 16 blocks (counters 128 bytes apart) and the last block of a group arrives at the top counter: 256
// atomics on ONE word took ~7 us per barrier (a device-scope atomic unit retires ~30 same-address operations per microsecond).
constexpr int kSgGroup = 16;
__device__ __forceinline__ bool sg_grid_barrier(unsigned *sync, unsigned G, unsigned &epoch, int *s_abort) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        const unsigned grp = blockIdx.x / kSgGroup, n_grp = (G + kSgGroup - 1) / kSgGroup;
        const unsigned gsize = min(static_cast<unsigned>(kSgGroup), G - grp * kSgGroup);
        const unsigned old = __hip_atomic_fetch_add(&sync[32 + 32 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old + 1u) % gsize == 0u) __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * n_grp;
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
    }
    __syncthreads();
    return *s_abort == 0;
}

// Near-tie replay inside the kernel for the one case that is frequent in RL-sized trees: a node of a few rows (an outlier split off by the
// levels above) whose best candidate sends every row to one side.  Its exact gain is 0; the reference's is score - parent, the SAME float32
// sums divided by sqrtf in one (math_ops.h:538-575) and by a double sqrt in the other (:504-524) -- the last bit decides between "split"
// and "leaf" (fitter.cpp:357).  One wave evaluates both from the node's rows in ascending order, operation for operation what
// k_near_replay (neartie.hip) does for this node in the level loop; returns fma(score, w, -parent).  Cosine only (under L2 both are the same
// operations and the gain is exactly 0).  scratch: this wave's slice of the (idle) histogram region, sg_tiny_words(n, D) words.
constexpr int kSgTinyRows = 16;
__device__ __forceinline__ int sg_tiny_words(int n, int D) { return kSgTinyRows + D + n * D; }   // scratch: rows | mean | the rows' gradients
__device__ float sg_tiny_zero_gain(const uint16_t *rc16, int N, int k, int n, const float *__restrict__ grads, int D, float w, uint32_t *scratch) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & (kWave - 1);
    int *rows = reinterpret_cast<int *>(scratch);
    float *mean = reinterpret_cast<float *>(scratch + kSgTinyRows), *g = mean + D;
    int found = 0;
    for (int r0 = 0; r0 < N && found < n; r0 += kWave) {
        const int r = r0 + lane;
        const bool in = r < N && rc16[2 * r] == static_cast<uint16_t>(k);
        const unsigned long long m = __ballot(in);
        if (in) { const int pos = found + __popcll(m & ((1ull << lane) - 1ull)); if (pos < kSgTinyRows) rows[pos] = r; }
        found += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < n * D; e += kWave) { const int i = e / D; g[e] = grads[static_cast<size_t>(rows[i]) * D + (e - i * D)]; }   // one round trip for all of them
    __builtin_amdgcn_wave_barrier();
    const float nf = static_cast<float>(n), rec = 1.0f / nf;
    for (int d = lane; d < D; d += kWave) {
        float sum = 0.0f;
        for (int i = 0; i < n; ++i) sum += g[i * D + d];
        mean[d] = sum * rec;
    }
    __builtin_amdgcn_wave_barrier();
    float gain = 0.0f;
    if (lane == 0) {
        const int D4 = D & ~3;
        float num = 0.0f, norm = 0.0f;
        for (int i = 0; i < n; ++i) {
            const float *gi = g + i * D;
            for (int c = 0; c < D4; ++c) { const float pr = gi[c] * mean[c]; num = num + pr; }
            for (int c = D4; c < D; ++c) num = fmaf(gi[c], mean[c], num);
        }
        for (int c = 0; c < D4; ++c) { const float pr = mean[c] * mean[c]; norm = norm + pr; }
        for (int c = D4; c < D; ++c) norm = fmaf(mean[c], mean[c], norm);
        const float den = norm * nf;
        const float score = den == 0.0f ? 0.0f : num / sqrtf(den);
        const float parent = den == 0.0f ? 0.0f : static_cast<float>(static_cast<double>(num) / sqrt(static_cast<double>(den)));
        gain = fmaf(score, w, -parent);
    }
    return __shfl(gain, 0, kWave);
}

// The in-kernel near-tie replay of ONE node of a greedy level (section "RL-sized steps" of DESIGN 3a): after the level's second scoring pass has
// collected this block's candidates inside the window (s_clist).  Returns 0: win[kf] holds the
// reference's choice; 1: a grid barrier gave up; 2: too many candidates -- the level loop takes the tree.
template <typename ACC>
__device__ __forceinline__ int sg_replay_node(const SmallGrowArgs &a, int level, bool obl, int n_act, unsigned &epoch, ACC *hist, const uint16_t *rc16, const int *tn,
                                                        SgBest *win, int *wcat, float *wthr, const SgCand *s_clist, const int *s_kf_p, const int *s_ccount_p,
                                                        const int *s_cover_p, int *s_near_p, int *s_abort_p) {
    const int N = a.N, D = a.D, B = a.B, G = a.G;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave, blk = blockIdx.x;
        const int kf = *s_kf_p;
        const int my_n = *s_cover_p ? 0xffff : min(*s_ccount_p, kSgCandCap);
        if (tid < kSgCandCap && tid < (my_n & 0x7fff) && my_n != 0xffff) {
            const SgCand cnd = s_clist[tid];
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(&a.nt.list[static_cast<size_t>(blk) * kSgCandCap + tid]);
            __hip_atomic_store(dst, (static_cast<unsigned long long>(static_cast<uint32_t>(cnd.ref)) << 32) | __float_as_uint(cnd.gain), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, (static_cast<unsigned long long>(cnd.nr) << 32) | cnd.slotbin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) __hip_atomic_store(&a.nt.count[blk], static_cast<uint32_t>(my_n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!sg_grid_barrier(a.sync, static_cast<unsigned>(G), epoch, s_abort_p)) return 1;
        // the idle histogram region: [replay core scratch | 1024 prefix words | merged list | final list | flags]
        uint32_t *scr = reinterpret_cast<uint32_t *>(hist);
        uint32_t *pre = scr + near_core_words(D, a.near_tile);
        SgCand *mlist = reinterpret_cast<SgCand *>(pre + 64);
        SgCand *fin = mlist + kSgMergeCap;
        int *st = reinterpret_cast<int *>(fin + kNearCands);     // [0] merged count (-1: too many), [1] classes
        if (wave == 0) {
            int total = 0;
            bool over = false;
            for (int b0 = 0; b0 < G; b0 += kWave) {
                const int b = b0 + lane;
                const uint32_t cnt = b < G ? __hip_atomic_load(&a.nt.count[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                over = over || __any(cnt == 0xffffu);
                int incl = static_cast<int>(cnt == 0xffffu ? 0u : cnt);
                for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o, kWave); if (lane >= o) incl += up; }
                const int mine_n = static_cast<int>(cnt == 0xffffu ? 0u : cnt), off = total + incl - mine_n;
                for (int e = 0; e < mine_n; ++e) {
                    if (off + e >= kSgMergeCap) { over = true; break; }
                    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&a.nt.list[static_cast<size_t>(b) * kSgCandCap + e]);
                    const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mlist[off + e] = SgCand{__uint_as_float(static_cast<uint32_t>(lo)), static_cast<int32_t>(static_cast<uint32_t>(lo >> 32)), static_cast<uint32_t>(hi), static_cast<uint32_t>(hi >> 32)};
                }
                total += __shfl(incl, kWave - 1, kWave);
                over = __any(over);
            }
            if (total > kSgMergeCap) over = true;
            __builtin_amdgcn_wave_barrier();
            // classes = distinct gains, each by its lowest reference index, best first (what k_near_list does in the level loop)
            int n_cls = 0;
            if (!over) {
                SgCand me{-INFINITY, 0x7fffffff, 0u, 0u};
                bool uniq = false;
                if (lane < total) {      // (batches of <= 8192 rows: classes by the gain alone, score_common.h near_class)
                    me = mlist[lane];
                    uniq = true;
                    for (int j = 0; j < total; ++j) { const SgCand o = mlist[j]; if (o.gain == me.gain && o.ref < me.ref) uniq = false; }
                }
                const unsigned long long um = __ballot(uniq);      // (at most 64 merged candidates: one per lane)
                int pos = 0;
                for (unsigned long long rest = um; rest; rest &= rest - 1) { const int j = __ffsll(static_cast<long long>(rest)) - 1; if (mlist[j].gain > me.gain) ++pos; }
                if (uniq && pos < kNearCands) fin[pos] = me;
                n_cls = min(kNearCands, static_cast<int>(__popcll(um)));
            }
            if (lane == 0) { st[0] = over ? -1 : total; st[1] = n_cls; }
        }
        __syncthreads();
        if (st[0] <= 0) return 2;      // too many candidates (or none): the level loop takes the tree
        const int n_cls = st[1];
        // greedy: the node's classes and its parent score; oblivious: every class on every node of the level (fitter.cpp:426-435 sums them)
        const int n_items = obl ? n_act * n_cls : n_cls + (level > 0 ? 1 : 0);
        const NearGrads ng{a.grads, a.meanden, D};
        for (int item = blk; item < n_items; item += G) {
            const bool is_parent = !obl && item == n_cls;
            const int inode = obl ? item / n_cls : kf, icls = obl ? item - inode * n_cls : item;
            const SgCand cnd = is_parent ? SgCand{0.0f, 0, 0u, 0u} : fin[icls];
            const int cslot = static_cast<int>(cnd.slotbin >> 16), cbin = static_cast<int>(cnd.slotbin & 0xffffu);
            const int ccat = is_parent ? 0 : a.slots[cslot].is_cat;
            const bool fm = a.codes_fm != nullptr && cslot < a.n_fm;
            const uint16_t *cs = fm ? a.codes_fm + static_cast<size_t>(cslot) * N : a.codes + (static_cast<size_t>(cslot >> 4) * N) * kCodeGroup + (cslot & (kCodeGroup - 1));
            const int cstride = fm ? 1 : kCodeGroup;
            // the node's rows in ascending order, bit 31 = goes right: thread t owns the rows [t * per, (t + 1) * per)
            int32_t *ent = a.nt.ent + static_cast<size_t>(blk) * N;
            const int per = (N + kSgThreads - 1) / kSgThreads, r_lo = min(N, tid * per), r_hi = min(N, r_lo + per);
            int cnt = 0, cntr = 0;      // this thread's rows of the node, and those of them that go right
            for (int r = r_lo; r < r_hi; ++r) {
                if (rc16[2 * r] != static_cast<uint16_t>(inode)) continue;
                ++cnt;
                if (!is_parent) { const int code = cs[static_cast<size_t>(r) * cstride]; cntr += (ccat ? (code == cbin) : (code > cbin)) ? 1 : 0; }
            }
            int incl = cnt, inclr = cntr;
            for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(incl, o, kWave), upr = __shfl_up(inclr, o, kWave); if (lane >= o) { incl += up; inclr += upr; } }
            __syncthreads();
            if (lane == kWave - 1) { pre[wave] = static_cast<uint32_t>(incl); pre[kSgWaves + wave] = static_cast<uint32_t>(inclr); }
            __syncthreads();
            int base = incl - cnt, n_node = 0, n_right = 0;
            for (int w = 0; w < kSgWaves; ++w) { if (w < wave) base += static_cast<int>(pre[w]); n_node += static_cast<int>(pre[w]); n_right += static_cast<int>(pre[kSgWaves + w]); }
            for (int r = r_lo; r < r_hi; ++r) {
                if (rc16[2 * r] != static_cast<uint16_t>(inode)) continue;
                bool right = false;
                if (!is_parent) { const int code = cs[static_cast<size_t>(r) * cstride]; right = ccat ? (code == cbin) : (code > cbin); }
                ent[base++] = r | static_cast<int32_t>(right ? 0x80000000u : 0u);
            }
            __threadfence_block();
            __syncthreads();
            const float res = near_replay_core(ent, n_node, n_right, ng, a.cosine != 0, is_parent, scr, a.near_tile);
            if (tid == 0) __hip_atomic_store(&a.nt.rep[is_parent ? kNearCands : (obl ? inode * kNearCands + icls : icls)], res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
        if (!sg_grid_barrier(a.sync, static_cast<unsigned>(G), epoch, s_abort_p)) return 1;
        // the reference's comparison of the replayed candidates (fitter.cpp:332-357): highest gain, first index among equals
        if (wave == 0) {
            SgBest m{-INFINITY, 0x7fffffff, 0u, 0u};
            const float par = (!obl && level > 0) ? __hip_atomic_load(&a.nt.rep[kNearCands], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            if (obl && lane < n_cls) {      // (sum over the level's nodes in node order, float32) * w  (fitter.cpp:426-435)
                const SgCand cnd = fin[lane];
                const int cslot = static_cast<int>(cnd.slotbin >> 16), cbin = static_cast<int>(cnd.slotbin & 0xffffu);
                const FeatureSlot csl = a.slots[cslot];
                const float wgt = a.cand_w[csl.cand_base + (csl.is_cat ? cbin - 1 : cbin)];
                float sc = 0.0f;
                for (int nd = 0; nd < n_act; ++nd) sc += __hip_atomic_load(&a.nt.rep[nd * kNearCands + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                m = SgBest{sc * wgt, cnd.ref, cnd.slotbin, 0u};
            } else if (lane < n_cls) {
                const SgCand cnd = fin[lane];
                const int cslot = static_cast<int>(cnd.slotbin >> 16), cbin = static_cast<int>(cnd.slotbin & 0xffffu);
                const FeatureSlot csl = a.slots[cslot];
                const float wgt = a.cand_w[csl.cand_base + (csl.is_cat ? cbin - 1 : cbin)];
                const float sc = __hip_atomic_load(&a.nt.rep[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                m = SgBest{fmaf(sc, wgt, -par), cnd.ref, cnd.slotbin, cnd.nr};
            }
            m = sg_wave_best(sg_better(SgBest{-INFINITY, 0x7fffffff, 0u, 0u}, m));
            if (lane == 0) {
                const int wslot = m.ref == 0x7fffffff ? 0 : static_cast<int>(m.slotbin >> 16);
                const int is_cat = a.slots[wslot].is_cat;
                const int tslot = wslot < a.n_thr_slots ? wslot : 0;
                const float tv = a.n_thr_slots > 0 ? a.thr[static_cast<size_t>(tslot) * B + min(static_cast<int>(m.slotbin & 0xffffu), B - 1)] : 0.0f;
                win[kf] = m;
                wcat[kf] = m.ref == 0x7fffffff ? 0 : is_cat;
                wthr[kf] = (m.ref == 0x7fffffff || is_cat) ? (is_cat ? INFINITY : 0.0f) : tv;
                *s_near_p = 0;
            }
        }
    return 0;
}

// REPLAY: the variant that replays a flagged node itself (sg_replay_node).  The default variant only detects and gives a flagged tree up
// (status 2); the host then launches this one for that tree -- with the replay inlined, the register allocation of the hot loops spills twice
// as much (every step lost 10 %), and as a called function it needs a 1.3 KB stack per lane (the launch alone took 0.15 ms longer).
template <typename ACC, bool REPLAY>
__global__ __launch_bounds__(kSgThreads) void k_small_grow(const SmallGrowArgs a) {
    extern __shared__ __align__(16) unsigned char sg_lds[];
    const SgLayout &L = a.L;
    uint32_t *rc = reinterpret_cast<uint32_t *>(sg_lds + L.rc);         // [N] per row: class code of the loaded slot << 16 | node word
    uint16_t *rc16 = reinterpret_cast<uint16_t *>(rc);                   //   node word = rc16[2 r]: index into the level's table, or kLeafBit | node id
    float *sw = reinterpret_cast<float *>(sg_lds + L.sw);               // [NB] candidate weights of the loaded slot
    int32_t *sref = reinterpret_cast<int32_t *>(sg_lds + L.sref);       // [NB] their reference indices
    int *tn_b = reinterpret_cast<int *>(sg_lds + L.tn);                 // [2][NC] rows of the level's nodes
    int *tid_b = reinterpret_cast<int *>(sg_lds + L.tid);               // [2][NC] their node ids (host numbering)
    uint32_t *psb_b = reinterpret_cast<uint32_t *>(sg_lds + L.psb);     // [2][NC][MD] path: slot << 16 | bin
    float *pv_b = reinterpret_cast<float *>(sg_lds + L.pv);             // [2][NC][MD] path: threshold value
    int *split = reinterpret_cast<int *>(sg_lds + L.split);             // [NC]
    int *cidx = reinterpret_cast<int *>(sg_lds + L.cidx);               // [NC][2] child's index in the next table, -1: not active
    int *cid = reinterpret_cast<int *>(sg_lds + L.cid);                 // [NC] left child's node id
    SgBest *win = reinterpret_cast<SgBest *>(sg_lds + L.win);           // [NC] the level's winners
    int *wcat = reinterpret_cast<int *>(sg_lds + L.wcat);               // [NC] winner slot is categorical
    float *wthr = reinterpret_cast<float *>(sg_lds + L.wthr);           // [NC] winner's threshold value (+inf: categorical)
    SgBest *nbest = reinterpret_cast<SgBest *>(sg_lds + L.nbest);       // [NC] greedy: this block's best per node
    float *nbest2 = reinterpret_cast<float *>(sg_lds + L.nbest2);       // [NC] ... and the best gain strictly below it
    float *npar = reinterpret_cast<float *>(sg_lds + L.npar);           // [NC] the node's parent score
    float *ssum = reinterpret_cast<float *>(sg_lds + L.ssum);           // [NB] oblivious: per candidate, sum over nodes
    unsigned char *leafflag = sg_lds + L.leafflag;                      // [NIDS]
    ACC *ttot = reinterpret_cast<ACC *>(sg_lds + L.ttot);               // [nb][T][W] tile totals, then carries
    long long *total = reinterpret_cast<long long *>(sg_lds + L.total); // [nb][W]
    double *total_f = reinterpret_cast<double *>(sg_lds + L.totalf);    // [nb][W]
    SgBest *ibest = reinterpret_cast<SgBest *>(sg_lds + L.ibest);       // [nb][T]
    float *ibest2 = reinterpret_cast<float *>(sg_lds + L.ibest2);       // [nb][T]
    SgBest *wbest = reinterpret_cast<SgBest *>(sg_lds + L.wbest);       // [16]
    float *wbest2 = reinterpret_cast<float *>(sg_lds + L.wbest2);       // [16]
    unsigned long long *lacc = reinterpret_cast<unsigned long long *>(sg_lds + L.lacc);   // [W]
    ACC *hist = reinterpret_cast<ACC *>(sg_lds + L.hist);               // [nb][NBe][W]

    __shared__ int s_abort, s_nact, s_nextid, s_nact_next, s_stop, s_near, s_kf, s_ccount, s_cover;
    __shared__ float s_lo;
    __shared__ SgCand s_clist[kSgCandCap];
    __shared__ SgBest s_bbest;
    __shared__ float s_bbest2;

    const int N = a.N, D = a.D, W = a.D + 1, B = a.B, MD = a.MD, NC = a.NC, G = a.G;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave, blk = blockIdx.x;
    const double inv_scale = a.scales->inv_scale;
    const bool obl = a.oblivious != 0;
    const bool detect = a.near_rel > 0.0f;
    unsigned epoch = 0;
    // measurement (a.prof): thread 0 of block 0 charges the time since the last mark to a phase (marks sit behind barriers)
    long long pt = a.prof ? wall_clock64() : 0;
    unsigned pacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pacc[i] = 0;
#define SG_MARK(i) do { if (a.prof && blk == 0 && tid == 0) { const long long n_ = wall_clock64(); pacc[i] += static_cast<unsigned>(n_ - pt); pt = n_; } } while (0)

    int cur = 0, level = 0, loaded_slot = -1;
    bool resumed = false;
    if (REPLAY && a.resume) {
        // continue the tree the default variant gave up at a flagged level: every block takes over its tables, row words and that level's
        // winners (the state is the same in all blocks; block 0 of the other kernel wrote it), and enters the level's second pass
        uint32_t *lw = reinterpret_cast<uint32_t *>(sg_lds);
        for (int i = tid; i < a.ckpt_words; i += kSgThreads) lw[i] = a.ckpt[16 + i];
        if (tid == 0) {
            s_abort = 0; s_stop = 0; s_ccount = 0; s_cover = 0;
            s_nact = static_cast<int>(a.ckpt[0]); s_nextid = static_cast<int>(a.ckpt[1]); s_kf = static_cast<int>(a.ckpt[4]); s_lo = __uint_as_float(a.ckpt[5]); s_near = 1;
        }
        level = static_cast<int>(a.ckpt[2]);
        cur = static_cast<int>(a.ckpt[3]);
        resumed = true;
    } else {
        for (int r = tid; r < N; r += kSgThreads) rc[r] = 0;
        for (int i = tid; i < a.NIDS; i += kSgThreads) leafflag[i] = 0;
        if (tid == 0) { tn_b[0] = N; tid_b[0] = 0; s_abort = 0; s_nact = 1; s_nextid = 1; s_stop = 0; s_near = 0; s_kf = 0; s_ccount = 0; s_cover = 0; s_lo = 0.0f; }
    }
    __syncthreads();

    bool ok = true, near_exit = false;
    unsigned n_replayed = 0;      // levels whose near-tie this launch replayed itself
    for (; level < MD; ++level) {
        const int n_act = s_nact;
        int *tn = tn_b + cur * NC, *tidc = tid_b + cur * NC;
        uint32_t *psb = psb_b + static_cast<size_t>(cur) * NC * MD;
        float *pv = pv_b + static_cast<size_t>(cur) * NC * MD;
        // ---- this block's slots ------------------------------------------------------------------------------------------------
        // (pass 1 only when the selection flags a near-tie at ONE node of a greedy level: the node's candidates are scored once more, those
        //  inside the window collected, re-scored in the reference's float32 order by the blocks, and the winner replaced -- below)
        bool leave = false;
        for (int pass = (REPLAY && resumed) ? 1 : 0; pass < (REPLAY ? 2 : 1) && !leave; ++pass) {
        resumed = false;
        const bool redo = REPLAY && pass == 1;
        if (redo) { if (tid == 0) { s_ccount = 0; s_cover = 0; } }
        else if (obl) { if (tid == 0) { s_bbest = SgBest{-INFINITY, 0x7fffffff, 0u, 0u}; s_bbest2 = -INFINITY; } }
        else for (int k = tid; k < n_act; k += kSgThreads) { nbest[k] = SgBest{-INFINITY, 0x7fffffff, 0u, 0u}; nbest2[k] = -INFINITY; npar[k] = 0.0f; }
        __syncthreads();
        const int k_begin = (redo && !obl) ? s_kf : 0, k_end = (redo && !obl) ? s_kf + 1 : n_act;    // (an oblivious level is scored again on every node)
        for (int fs = blk; fs < a.n_slots; fs += G) {
            const FeatureSlot sl = a.slots[fs];
            const int NBe = sl.n_cand + 1;                       // classes of this slot
            const int T = (NBe - 1 + kWave - 1) / kWave;         // 64-class tiles over the classes 1 .. NBe-1
            if (fs != loaded_slot) {                             // the slot's class codes in row order, its candidates' weights and reference indices
                const bool fm = a.codes_fm != nullptr && fs < a.n_fm;
                const uint16_t *cs = fm ? a.codes_fm + static_cast<size_t>(fs) * N : a.codes + (static_cast<size_t>(fs >> 4) * N) * kCodeGroup + (fs & (kCodeGroup - 1));
                const int cstride = fm ? 1 : kCodeGroup;
                for (int r0 = tid; r0 < N; r0 += kSgThreads * 8) {
                    uint16_t cv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cv[u] = cs[static_cast<size_t>(min(r0 + u * kSgThreads, N - 1)) * cstride];
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (r0 + u * kSgThreads < N) rc16[2 * (r0 + u * kSgThreads) + 1] = cv[u];
                }
                for (int k = tid; k < sl.n_cand; k += kSgThreads) { sw[k] = a.cand_w[sl.cand_base + k]; sref[k] = a.cand_ref[sl.cand_base + k]; }
                loaded_slot = fs;
            }
            SG_MARK(0);
            if (obl) for (int k = tid; k < sl.n_cand; k += kSgThreads) ssum[k] = 0.0f;
            for (int k0 = k_begin; k0 < k_end; k0 += a.nb_cap) {
                const int nbk = min(a.nb_cap, k_end - k0);
                const int hwords = nbk * NBe * W;
                {   // (16-byte stores; the region's capacity is a multiple of 16 bytes)
                    typedef unsigned int sg_u4 __attribute__((ext_vector_type(4)));
                    sg_u4 *hz = reinterpret_cast<sg_u4 *>(hist);
                    const int n16 = static_cast<int>((static_cast<size_t>(hwords) * sizeof(ACC) + 15) / 16);
                    for (int i = tid; i < n16; i += kSgThreads) hz[i] = sg_u4{0u, 0u, 0u, 0u};
                }
                __syncthreads();
                SG_MARK(1);
                // -- accumulate: one ROW per lane (four rows per thread and round).  The row's class word is read once, its histogram base
                //    computed once, and every field is one ds_add with an immediate offset: ~2.5 VALU instructions per (row, field) -- the first
                //    version (one lane per (row, field)) spent 40 on index arithmetic and was bound by them (10 us per level at 4096 x 9).  The
                //    lanes of a wave hit random classes, which the LDS atomic unit absorbs at its issue rate (DESIGN, k_hist_build).
                if ((D & 3) == 0) sg_accumulate_rows<ACC, 4>(a.qg, rc, hist, N, D, NBe, k0, nbk, tid);
                else if ((D & 1) == 0) sg_accumulate_rows<ACC, 2>(a.qg, rc, hist, N, D, NBe, k0, nbk, tid);
                else sg_accumulate_rows<ACC, 1>(a.qg, rc, hist, N, D, NBe, k0, nbk, tid);
                __syncthreads();
                SG_MARK(2);
                // -- phase A: per (node, tile) suffix sums over the tile's classes (in place) and the tile totals
                const int n_items = nbk * T;
                for (int it = wave; it < n_items; it += kSgWaves) {
                    const int kb = it / T, t = it - kb * T;
                    const int c = NBe - 1 - (t * kWave + lane);   // this lane's class; the tile's lanes run DOWN the classes
                    const bool have = c >= 1;
                    ACC *hc = hist + (static_cast<size_t>(kb) * NBe + (have ? c : 0)) * W;
                    for (int w0 = 0; w0 < W; w0 += 9) {
                        ACC v[9];
#pragma unroll
                        for (int j = 0; j < 9; ++j) v[j] = (have && w0 + j < W) ? hc[min(w0 + j, W - 1)] : 0;
                        wave_scan9(v);
                        if (!sl.is_cat) {
#pragma unroll
                            for (int j = 0; j < 9; ++j) if (have && w0 + j < W) hc[w0 + j] = v[j];
                        }
                        if (lane == kWave - 1) {
#pragma unroll
                            for (int j = 0; j < 9; ++j) if (w0 + j < W) ttot[(static_cast<size_t>(kb) * T + t) * W + w0 + j] = v[j];
                        }
                    }
                }
                __syncthreads();
                SG_MARK(3);
                // -- carries (sum of the tiles above) and node totals (all tiles + class 0)
                for (int i = tid; i < nbk * W; i += kSgThreads) {
                    const int kb = i / W, j = i - kb * W;
                    long long run = 0;
                    for (int t = 0; t < T; ++t) {
                        ACC *p = &ttot[(static_cast<size_t>(kb) * T + t) * W + j];
                        const ACC x = *p;
                        *p = static_cast<ACC>(run);     // (a sum over a subset of the node's rows: fits the accumulator type)
                        run += x;
                    }
                    run += static_cast<long long>(hist[(static_cast<size_t>(kb) * NBe) * W + j]);
                    total[i] = run;
                    total_f[i] = static_cast<double>(run);
                }
                __syncthreads();
                SG_MARK(4);
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
                    const ACC *cy = ttot + (static_cast<size_t>(kb) * T + t) * W;
                    const bool scanned = !sl.is_cat;
                    float out = -INFINITY;
                    int n_r = 0;
                    if (have) {
                        n_r = static_cast<int>(hc[D] + (scanned ? cy[D] : 0));   // (row counts: at most 8192)
                        const int n_l = static_cast<int>(n_tot) - n_r;
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
                        if (!reject) {
                            if (scanned) out = candidate_score([&](int d) { return static_cast<double>(static_cast<ACC>(hc[d] + cy[d])); }, totf, D, n_l, n_r, a.cosine, inv_scale);
                            else out = candidate_score([&](int d) { return static_cast<double>(hc[d]); }, totf, D, n_l, n_r, a.cosine, inv_scale);
                        }
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
                            mine = SgBest{fmaf(out, sw[k], -par_sub), sref[k],
                                          (static_cast<uint32_t>(fs) << 16) | static_cast<uint32_t>(sl.is_cat ? k + 1 : k), static_cast<uint32_t>(n_r)};
                        }
                        if (REPLAY && redo) {
                            // collect the candidates inside the window; of a run of neighbours with the same gain (thresholds between the same
                            // two rows of the node) only the lowest reference index -- the last lane of the run
                            const float below = __shfl_down(mine.v, 1, kWave);
                            const bool keep = have && mine.v >= s_lo && (lane == kWave - 1 || below != mine.v);
                            const unsigned long long mk = __ballot(keep);
                            if (mk) {
                                int base = 0;
                                if (lane == 0) base = atomicAdd(&s_ccount, __popcll(mk));
                                base = __builtin_amdgcn_readfirstlane(base);
                                if (keep) {
                                    const int at = base + __popcll(mk & ((1ull << lane) - 1ull));
                                    if (at < kSgCandCap) s_clist[at] = SgCand{mine.v, mine.ref, mine.slotbin, mine.pad};
                                    else s_cover = 1;
                                }
                            }
                            continue;
                        }
                        const float own = mine.v;
                        mine = sg_wave_best(sg_better(SgBest{-INFINITY, 0x7fffffff, 0u, 0u}, mine));
                        const float sec = detect ? sg_wave_second(own, -INFINITY, mine.v) : -INFINITY;
                        if (lane == 0) { ibest[it] = mine; ibest2[it] = sec; if (t == 0) npar[k_abs] = par_sub; }
                    }
                }
                __syncthreads();
                SG_MARK(5);
                // -- phase C
                if (obl) {
                    for (int k = tid; k < sl.n_cand; k += kSgThreads) {   // fp32 sum over the level's nodes in node order (fitter.cpp:426-435)
                        float s = ssum[k];
                        for (int kb = 0; kb < nbk; ++kb) s += __int_as_float(static_cast<int>(hist[(static_cast<size_t>(kb) * NBe + k + 1) * W]));
                        ssum[k] = s;
                    }
                } else if (!redo) {
                    for (int kb = tid; kb < nbk; kb += kSgThreads) {
                        SgBest b = nbest[k0 + kb];
                        float b2 = nbest2[k0 + kb];
                        for (int t = 0; t < T; ++t) { b2 = second_distinct(b.v, b2, ibest[kb * T + t].v, ibest2[kb * T + t]); b = sg_better(b, ibest[kb * T + t]); }
                        nbest[k0 + kb] = b;
                        nbest2[k0 + kb] = b2;
                    }
                }
                __syncthreads();
                SG_MARK(6);
            }
            if (REPLAY && redo && obl) {
                // collect the slot's candidates inside the window; of neighbours with the same score only the lowest reference index
                for (int k0c = 0; k0c < sl.n_cand; k0c += kSgThreads) {
                    const int k = k0c + tid;
                    const float sc = k < sl.n_cand ? ssum[k] * sw[k] : -INFINITY;
                    const bool keep = k < sl.n_cand && sc >= s_lo && (k == 0 || ssum[k - 1] * sw[k - 1] != sc);
                    const unsigned long long mk = __ballot(keep);
                    if (mk) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&s_ccount, __popcll(mk));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (keep) {
                            const int at = base + __popcll(mk & ((1ull << lane) - 1ull));
                            if (at < kSgCandCap) s_clist[at] = SgCand{sc, sref[k], (static_cast<uint32_t>(fs) << 16) | static_cast<uint32_t>(sl.is_cat ? k + 1 : k), 0u};
                            else s_cover = 1;
                        }
                    }
                }
                __syncthreads();
            } else if (obl) {   // the slot's best candidate: (sum over nodes) * w, lowest reference index among maxima (fitter.cpp:435-444)
                SgBest mine{-INFINITY, 0x7fffffff, 0u, 0u};
                float sec = -INFINITY;
                for (int k = tid; k < sl.n_cand; k += kSgThreads) {
                    const float sc = ssum[k] * sw[k];
                    sec = second_distinct(mine.v, sec, sc, -INFINITY);
                    mine = sg_better(mine, SgBest{sc, sref[k], (static_cast<uint32_t>(fs) << 16) | static_cast<uint32_t>(sl.is_cat ? k + 1 : k), 0u});
                }
                const float own = mine.v;
                mine = sg_wave_best(mine);
                if (detect) sec = sg_wave_second(own, sec, mine.v);
                if (lane == 0) { wbest[wave] = mine; wbest2[wave] = sec; }
                __syncthreads();
                if (wave == 0) {     // (the next write of wbest sits behind the barriers of the next slot / level)
                    SgBest b = lane < kSgWaves ? wbest[lane] : SgBest{-INFINITY, 0x7fffffff, 0u, 0u};
                    float b2 = lane < kSgWaves ? wbest2[lane] : -INFINITY;
                    const float own = b.v;
                    b = sg_wave_best(b);
                    if (detect) b2 = sg_wave_second(own, b2, b.v);
                    if (lane == 0) { s_bbest2 = second_distinct(s_bbest.v, s_bbest2, b.v, b2); s_bbest = sg_better(s_bbest, b); }
                }
            }
        }
        __syncthreads();
        if constexpr (REPLAY) if (redo) {
            const int rr = sg_replay_node<ACC>(a, level, obl, n_act, epoch, hist, rc16, tn, win, wcat, wthr, s_clist, &s_kf, &s_ccount, &s_cover, &s_near, &s_abort);
            if (rr == 1) { ok = false; leave = true; break; }
            if (rr == 2) { near_exit = true; leave = true; break; }
            ++n_replayed;
            __syncthreads();
            break;
        }
        // ---- publish this block's bests, ONE grid barrier, reduce to the winners -------------------------------------------------
        SG_MARK(7);
        SgBest *lv = a.bests + static_cast<size_t>(level) * (obl ? 1 : NC) * G;
        // Near-tie detection travels in the record: bit 31 of `pad` = "this block's own runner-up is inside the window of its best".  The level's
        // runner-up is either another block's best (all of them are read below anyway) or the runner-up of a block whose best IS the level's
        // best -- and for those the block's own test is the level's test (same best, same parent score).
        if (obl) {
            if (tid == 0) {
                SgBest r = s_bbest;
                if (detect && s_bbest2 != -INFINITY && r.v - s_bbest2 <= a.near_rel * fabsf(r.v)) r.pad |= 0x80000000u;
                sg_store_best(&lv[blk], r);
            }
        } else for (int k = tid; k < n_act; k += kSgThreads) {
            SgBest r = nbest[k];
            if (detect && nbest2[k] != -INFINITY && r.v - nbest2[k] <= a.near_rel * fmaxf(fabsf(r.v + npar[k]), fabsf(npar[k]))) r.pad |= 0x80000000u;
            sg_store_best(&lv[static_cast<size_t>(k) * G + blk], r);
        }
        if (!sg_grid_barrier(a.sync, static_cast<unsigned>(G), epoch, &s_abort)) { ok = false; leave = true; break; }
        SG_MARK(8);
        const int n_win = obl ? 1 : n_act;
        for (int k = wave; k < n_win; k += kSgWaves) {
            SgBest m{-INFINITY, 0x7fffffff, 0u, 0u};
            float m2 = -INFINITY;
            bool mf = false;        // a record with this lane's best value carries the flag
            for (int q0 = lane; q0 < G; q0 += kWave * 4) {          // (G <= 256 blocks: one round of four loads in flight)
                SgBest b4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b4[u] = sg_load_best(lv + static_cast<size_t>(k) * G + min(q0 + u * kWave, G - 1));
#pragma unroll
                for (int u = 0; u < 4; ++u) if (q0 + u * kWave < G) {
                    const bool f = (b4[u].pad & 0x80000000u) != 0;
                    if (b4[u].v > m.v) mf = f; else if (b4[u].v == m.v) mf = mf || f;
                    m2 = second_distinct(m.v, m2, b4[u].v, -INFINITY);
                    m = sg_better(m, b4[u]);
                }
            }
            const float own = m.v;
            m = sg_wave_best(m);
            m.pad &= 0x7fffffffu;
            bool flagged = false;
            if (detect) { m2 = sg_wave_second(own, m2, m.v); flagged = __any(mf && own == m.v) != 0; }
            const float par = (detect && !obl) ? npar[k] : 0.0f;       // (every block derives the same parent score from the node's totals)
            int tiny_k = 0;
            if (lane == 0 && detect && m.v != -INFINITY) {
                // the same test as k_resolve_splits: the runner-up within near_rel of the winner (relative to the scores' magnitude), or a greedy
                // gain that close to zero -- except, under L2, a zero gain whose winner sends every row to one side (exactly 0 in the reference
                // too).  Every block evaluates it on the same records, so all of them leave together.
                const float mag = obl ? fabsf(m.v) : fmaxf(fabsf(m.v + par), fabsf(par));
                const float win_w = a.near_rel * mag;
                bool near = flagged || (m2 != -INFINITY && m.v - m2 <= win_w);
                int tiny = 0;
                if (!obl && level > 0 && fabsf(m.v) <= win_w) {
                    const int nr = static_cast<int>(m.pad), nn = tn[k];
                    const bool one_sided = nr == 0 || nr == nn;
                    if (!one_sided) near = true;
                    else if (a.cosine) { if (nn <= kSgTinyRows && sg_tiny_words(nn, D) <= a.tiny_words) tiny = 1; else near = true; }
                }
                if (near) { atomicAdd(&s_near, 1); s_kf = k; s_lo = m.v - win_w; }
                else if (tiny) tiny_k = 1;
            }
            tiny_k = __builtin_amdgcn_readfirstlane(tiny_k);
            if (tiny_k) {     // (wave-uniform) the zero gain of a node of a few rows, settled here: this wave's slice of the idle histogram region is its scratch
                const int wslot = static_cast<int>(m.slotbin >> 16), wbin = static_cast<int>(m.slotbin & 0xffffu);
                const FeatureSlot wsl = a.slots[wslot];
                const float wgt = a.cand_w[wsl.cand_base + (wsl.is_cat ? wbin - 1 : wbin)];
                m.v = sg_tiny_zero_gain(rc16, N, k, tn[k], a.grads, D, wgt, reinterpret_cast<uint32_t *>(hist) + static_cast<size_t>(wave) * a.tiny_words);
            }
            if (lane == 0) {
                // the winner's kind and threshold value: both loads issued here, together (the paths of the children and the result block
                // need the value; fetched there it was a second DRAM round trip behind the routing pass's)
                const int wslot = m.ref == 0x7fffffff ? 0 : static_cast<int>(m.slotbin >> 16);
                const int is_cat = a.slots[wslot].is_cat;
                const int tslot = wslot < a.n_thr_slots ? wslot : 0;
                const float tv = a.n_thr_slots > 0 ? a.thr[static_cast<size_t>(tslot) * B + min(static_cast<int>(m.slotbin & 0xffffu), B - 1)] : 0.0f;
                win[k] = m;
                wcat[k] = m.ref == 0x7fffffff ? 0 : is_cat;
                wthr[k] = (m.ref == 0x7fffffff || is_cat) ? (is_cat ? INFINITY : 0.0f) : tv;
            }
        }
        __syncthreads();
        SG_MARK(9);
        if (s_near == 0) break;                                        // (the usual case: no second pass)
        if (!REPLAY && s_near == 1 && a.near_in_kernel && a.ckpt != nullptr && blk == 0) {
            // leave the tree's state for the variant that replays (it enters this level's second pass): header, then the fixed part of LDS
            const uint32_t *lw = reinterpret_cast<const uint32_t *>(sg_lds);
            for (int i = tid; i < a.ckpt_words; i += kSgThreads) a.ckpt[16 + i] = lw[i];
            if (tid == 0) {
                a.ckpt[0] = static_cast<uint32_t>(s_nact); a.ckpt[1] = static_cast<uint32_t>(s_nextid); a.ckpt[2] = static_cast<uint32_t>(level); a.ckpt[3] = static_cast<uint32_t>(cur);
                a.ckpt[4] = static_cast<uint32_t>(s_kf); a.ckpt[5] = __float_as_uint(s_lo);
            }
        }
        if (!REPLAY || s_near != 1 || !a.near_in_kernel) { near_exit = true; leave = true; break; }   // more than one node, or no room: the level loop takes the tree
        }   // passes
        if (leave) break;
        // ---- children: node ids in the host's order (for k in splitting: left, right), next level's table.  The child sizes of a greedy
        //      split travel with its winner (SgBest::pad); an oblivious level keeps both children of every node, empty or not, so its table
        //      needs no sizes (the host derives them from the leaves' row counts afterwards).
        const bool last_level = level + 1 == MD;
        int *tn2 = tn_b + (cur ^ 1) * NC, *tid2 = tid_b + (cur ^ 1) * NC;
        uint32_t *psb2 = psb_b + static_cast<size_t>(cur ^ 1) * NC * MD;
        float *pv2 = pv_b + static_cast<size_t>(cur ^ 1) * NC * MD;
        if (wave == 0) {
            int run_split = 0, run_act = 0;
            const int next_id = s_nextid;
            const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (kWave - lane));
            const float v0 = win[0].v;
            for (int base = 0; base < n_act; base += kWave) {
                const int k = base + lane;
                const bool valid = k < n_act;
                // fitter.cpp:458 oblivious: grow while any candidate is finite; fitter.cpp:357 greedy: split iff best >= 0
                const bool sp = valid && (obl ? (v0 != -INFINITY) : (win[k].v >= 0.0f));
                const unsigned long long m = __ballot(sp);
                const int my_id = next_id + 2 * (run_split + __popcll(m & lt));
                const int nr = (valid && !obl) ? static_cast<int>(win[k].pad) : 0, nl = (valid && !obl) ? tn[k] - nr : 0;
                const bool al = sp && !last_level && (obl || nl > 0), ar = sp && !last_level && (obl || nr > 0);
                const unsigned long long ml = __ballot(al), mr = __ballot(ar);
                const int il = run_act + __popcll(ml & lt) + __popcll(mr & lt), ir = il + (al ? 1 : 0);
                if (valid) {
                    split[k] = sp ? 1 : 0;
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
        SG_MARK(10);
        // ---- routing: every row of a splitting node moves to its child (its index in the next table, or the child's node id when the child
        //      is a leaf); rows of a node that did not split stay with it as a leaf.  ONE pass: the children's places are known (above).
        for (int r0 = tid; r0 < N; r0 += kSgThreads * 4) {       // four rows per thread and round: everything is loaded before anything is decided
            unsigned nv[4];
            int code[4], bin[4], wc[4], go[4][2], stay[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u * kSgThreads;
                nv[u] = r < N ? (rc[r] & 0xffffu) : 0xffffu;
                const int k = nv[u] < kLeafBit ? static_cast<int>(nv[u]) : 0;
                const SgBest w = win[obl ? 0 : k];
                const int slot = static_cast<int>(w.slotbin >> 16);
                bin[u] = static_cast<int>(w.slotbin & 0xffffu);
                wc[u] = wcat[obl ? 0 : k];
                const int rr = min(r, N - 1);
                const uint16_t *cp = (a.codes_fm != nullptr && slot < a.n_fm) ? a.codes_fm + static_cast<size_t>(slot) * N + rr
                                                                              : a.codes + (static_cast<size_t>(slot >> 4) * N + rr) * kCodeGroup + (slot & (kCodeGroup - 1));
                code[u] = *cp;   // (unconditional: a load in a branch is waited for at once)
                // where the row goes for either outcome (read while the code is on its way)
                const int c0 = cidx[2 * k], c1 = cidx[2 * k + 1], idl = cid[k];
                go[u][0] = c0 >= 0 ? c0 : static_cast<int>(kLeafBit | static_cast<unsigned>(idl));
                go[u][1] = c1 >= 0 ? c1 : static_cast<int>(kLeafBit | static_cast<unsigned>(idl + 1));
                stay[u] = split[k] ? -1 : static_cast<int>(kLeafBit | static_cast<unsigned>(tidc[k]));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (nv[u] >= kLeafBit) continue;
                const int r = r0 + u * kSgThreads;
                const bool right = wc[u] ? (code[u] == bin[u]) : (code[u] > bin[u]);
                const int w2 = stay[u] >= 0 ? stay[u] : (right ? go[u][1] : go[u][0]);
                rc16[2 * r] = static_cast<uint16_t>(w2);
            }
        }
        // paths of the active children: the parent's, plus the new condition
        for (int i = tid; i < n_act * 2 * (level + 1); i += kSgThreads) {
            const int p = i % (level + 1), ks = i / (level + 1), k = ks >> 1, side = ks & 1;
            const int dst = split[k] ? cidx[2 * k + side] : -1;
            if (dst < 0) continue;
            if (p < level) { psb2[dst * MD + p] = psb[k * MD + p]; pv2[dst * MD + p] = pv[k * MD + p]; }
            else {
                const SgBest w = win[obl ? 0 : k];
                psb2[dst * MD + p] = w.slotbin;
                pv2[dst * MD + p] = wthr[obl ? 0 : k];
            }
        }
        // block 0: the level's result block for the host (the layout digest_level reads) + the winners' threshold values
        if (blk == 0) {   // (staged in device memory: a store to the host's pinned block is a PCIe round trip in front of the next block barrier)
            char *res = a.res_dev + static_cast<size_t>(level) * a.res_stride;
            int32_t *r_idx = reinterpret_cast<int32_t *>(res);
            float *r_score = reinterpret_cast<float *>(res + 4 * static_cast<size_t>(a.max_front));
            int64_t *r_cnt = reinterpret_cast<int64_t *>(res + 8 * static_cast<size_t>(a.max_front));
            float *r_thr = reinterpret_cast<float *>(res + 40 * static_cast<size_t>(a.max_front));
            for (int k = tid; k < n_act; k += kSgThreads) {
                const SgBest w = win[obl ? 0 : k];
                if (!obl || k == 0) { r_idx[k] = w.ref == 0x7fffffff ? 0 : w.ref; r_score[k] = w.v; }
                r_cnt[k] = obl ? 0 : tn[k];                                            // (oblivious: the host derives the sizes from the leaves)
                r_cnt[static_cast<size_t>(a.max_front) + k] = obl ? 0 : static_cast<int>(w.pad);
                r_thr[k] = wcat[obl ? 0 : k] ? 0.0f : wthr[obl ? 0 : k];
            }
        }
        __syncthreads();
        SG_MARK(11);
        const int stop = s_stop;
        if (tid == 0) s_nact = s_nact_next;
        cur ^= 1;
        __syncthreads();
        SG_MARK(12);
        if (stop || last_level) { ++level; break; }
    }
    // Oblivious trees keep growing with EMPTY nodes in the table; whatever is still active when the loop ends is a leaf.
    if (ok && !near_exit) {
        const int n_act = s_nact;
        const int *tidc = tid_b + cur * NC;
        for (int k = tid; k < n_act; k += kSgThreads) leafflag[tidc[k]] = 1;
        for (int r = tid; r < N; r += kSgThreads) {
            const unsigned v = rc16[2 * r];
            if (!(v & kLeafBit)) rc16[2 * r] = static_cast<uint16_t>(kLeafBit | static_cast<unsigned>(tidc[v & 0x3fffu]));
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
                    if (rc16[2 * r] == want) {
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
    SG_MARK(13);
    if (a.prof && blk == 0 && tid == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a.prof[i] = pacc[i];
    }
#undef SG_MARK
    if (blk == 0) {   // the levels' result blocks, written by this block's own threads: out to the host in one go
        __syncthreads();
        const int words = level * a.res_stride / 4;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.res_dev);
        uint32_t *dst = reinterpret_cast<uint32_t *>(a.res);
        for (int i = tid; i < words; i += kSgThreads) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the last block to finish publishes the sequence word (and hands the counters back zeroed) --------------------------------
    __syncthreads();
    if (tid == 0) {
        if (blk == 0) {
            a.status[1] = static_cast<uint32_t>(level); a.status[2] = static_cast<uint32_t>(s_nextid); a.status[4] = n_replayed; a.status[5] = (!REPLAY && near_exit && s_near == 1 && a.near_in_kernel && a.ckpt != nullptr) ? 1u : 0u;
            if (a.scales_out) *a.scales_out = *a.scales;
        }
        __threadfence_system();
        if (atomicAdd(&a.sync[1], 1u) == static_cast<unsigned>(G) - 1u) {
            const unsigned aborted = __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.sync[0] = 0u; a.sync[1] = 0u; a.sync[2] = 0u;
            for (int g = 0; g * kSgGroup < G; ++g) a.sync[32 + 32 * g] = 0u;
            a.status[3] = aborted || !ok ? 1u : (near_exit ? 2u : 0u);
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
static size_t small_grow_rep_bytes(int MD) { return 256 * ((sizeof(float) * ((static_cast<size_t>(1) << std::max(0, MD - 1)) * kNearCands + kNearCands + 1) + 255) / 256); }
size_t small_grow_near_bytes(int G, int N, int MD) {   // (+ 192 KiB: the checkpoint -- at most the kernel's whole LDS)   // SgNearScratch: counts | candidate lists | replayed scores | ordered row lists
    return 256 * ((sizeof(uint32_t) * G + 255) / 256) + sizeof(SgCand) * static_cast<size_t>(G) * kSgCandCap + small_grow_rep_bytes(MD) + sizeof(int32_t) * static_cast<size_t>(G) * N + 192 * 1024;
}
size_t small_grow_res_stride(int MD) { return static_cast<size_t>(1 << std::max(0, MD - 1)) * 44 + 64; }

int small_grow_blocks(int n_slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    const int cap = [] { const char *e = hooks::raw(hooks::SMALL_GROW_BLOCKS); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 1 << 30; }();   /* read per call: the tests flip it */   // test / measurement hook
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
    L.rc = take(4 * N);
    L.sw = take(4 * NB);
    L.sref = take(4 * NB);
    L.tn = take(4 * 2 * NC);
    L.tid = take(4 * 2 * NC);
    L.psb = take(4 * 2 * NC * MD);
    L.pv = take(4 * 2 * NC * MD);
    L.split = take(4 * NC);
    L.cidx = take(8 * NC);
    L.cid = take(4 * NC);
    L.win = take(16 * NC);
    L.wcat = take(4 * NC);
    L.wthr = take(4 * NC);
    L.nbest = take(16 * NC);
    L.nbest2 = take(4 * NC);
    L.npar = take(4 * NC);
    L.ssum = take(4 * NB);
    L.leafflag = take(NIDS);
    L.wbest = take(16 * kSgWaves);
    L.wbest2 = take(4 * kSgWaves);
    L.lacc = take(8 * W);
    const int budget = 160 * 1024 - 1024 /* static __shared__ + slack */ - off - 8 * kWave - 16;
    const int per_node = NB * W * acc_bytes + Tmax * W * acc_bytes + W * 16 + Tmax * 20 + 80;
    nb_cap = budget > 0 ? std::min(NC, budget / per_node) : 0;
    if (nb_cap <= 0) { nb_cap = 0; return L; }
    L.ttot = take(acc_bytes * nb_cap * Tmax * W);
    L.total = take(8 * nb_cap * W);
    L.totalf = take(8 * nb_cap * W);
    L.ibest = take(16 * nb_cap * Tmax);
    L.ibest2 = take(4 * nb_cap * Tmax);
    (void)take(acc_bytes * kWave);                      // the accumulate loop's per-lane sink words sit right in front of the histogram
    L.hist = take(acc_bytes * nb_cap * NB * W);
    L.total_bytes = off;
    return L;
}
}  // namespace

// the kernel asks for (almost) the whole 160 KiB of a gfx950 CU; a device that offers less per block keeps to the level loop
static bool device_lds_fits() {
    static PerDeviceOnce once;
    static int ok[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (once.first()) {
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); optin = 0; }
        ok[dev] = (optin <= 0 || optin >= 160 * 1024 - 1024) ? 1 : 0;   // (unknown: let the launch decide -- a failed launch falls back too)
    }
    return ok[dev] != 0;
}

bool small_grow_supported(int N, int D, int NB, int MD, int n_slots, int n_cand) {
    if (!device_lds_fits()) return false;
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
    a.codes = io.codes; a.codes_fm = io.codes_fm; a.n_fm = io.n_fm; a.qg = io.qg; a.grads = io.grads; a.scales = io.scales; a.slots = io.slots; a.thr = io.thr; a.n_thr_slots = io.n_thr_slots; a.cand_w = io.cand_w; a.cand_ref = io.cand_ref;
    a.N = io.N; a.D = io.D; a.B = io.B; a.n_slots = io.n_slots; a.NB = io.NB; a.MD = io.MD; a.min_data = io.min_data; a.cosine = io.cosine ? 1 : 0; a.oblivious = io.oblivious ? 1 : 0;
    a.G = io.G; a.NC = 1 << std::max(0, io.MD - 1); a.NIDS = 2 << io.MD;
    a.magicW = static_cast<uint32_t>((1ull << 32) / static_cast<unsigned>(io.D + 1)) + 1u;
    a.bests = static_cast<SgBest *>(io.bests);
    a.near_rel = io.near_rel;
    a.meanden = io.meanden;
    {
        char *base = static_cast<char *>(io.near_scratch);
        const size_t o_list = 256 * ((sizeof(uint32_t) * io.G + 255) / 256), o_rep = o_list + sizeof(SgCand) * static_cast<size_t>(io.G) * kSgCandCap, o_ent = o_rep + small_grow_rep_bytes(io.MD);
        a.nt.count = reinterpret_cast<uint32_t *>(base);
        a.nt.list = reinterpret_cast<SgCand *>(base + o_list);
        a.nt.rep = reinterpret_cast<float *>(base + o_rep);
        a.nt.ent = reinterpret_cast<int32_t *>(base + o_ent);
        a.ckpt = base ? reinterpret_cast<uint32_t *>(base + o_ent + sizeof(int32_t) * static_cast<size_t>(io.G) * io.N) : nullptr;
        a.ckpt_words = a.L.ttot / 4;
        a.resume = (io.replay && io.resume) ? 1 : 0;
        const size_t hist_bytes = static_cast<size_t>(acc_bytes) * a.nb_cap * io.NB * (io.D + 1);
        const size_t fixed = sizeof(uint32_t) * (static_cast<size_t>(near_core_words(io.D, 0)) + 64) + sizeof(SgCand) * (kSgMergeCap + kNearCands) + 64;
        const int Dp = (io.D + 3) & ~3;
        a.near_tile = hist_bytes > fixed ? static_cast<int>(std::min<size_t>(kNearTile, ((hist_bytes - fixed) / 4) & ~static_cast<size_t>(255))) : 0;
        const size_t need = a.near_tile >= std::max(256, 8 * Dp) ? 0 : hist_bytes + 1;
        a.near_in_kernel = (base != nullptr && hist_bytes >= need && io.D <= kNearMaxD) ? 1 : 0;
    }
    a.tiny_words = static_cast<int>((static_cast<size_t>(acc_bytes) * a.nb_cap * io.NB * (io.D + 1) / 4 / kSgWaves) & ~static_cast<size_t>(3));
    a.sync = io.sync; a.res = io.res; a.res_dev = io.res_dev; a.res_stride = static_cast<int>(small_grow_res_stride(io.MD)); a.max_front = a.NC;
    a.acc = io.acc; a.status = io.status; a.seq = io.seq; a.prof = io.prof; a.scales_out = io.scales_out;
    static PerDeviceOnce attr[4];
    const int which = (acc_bytes == 4 ? 0 : 1) + (io.replay ? 2 : 0);
    const void *fn = which == 0 ? reinterpret_cast<const void *>(k_small_grow<int32_t, false>) : which == 1 ? reinterpret_cast<const void *>(k_small_grow<long long, false>)
                   : which == 2 ? reinterpret_cast<const void *>(k_small_grow<int32_t, true>) : reinterpret_cast<const void *>(k_small_grow<long long, true>);
    if (attr[which].first() && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) { (void)hipGetLastError(); attr[which].done = 0; return false; }
    if (which == 0) hipLaunchKernelGGL((k_small_grow<int32_t, false>), dim3(io.G), dim3(kSgThreads), a.L.total_bytes, s, a);
    else if (which == 1) hipLaunchKernelGGL((k_small_grow<long long, false>), dim3(io.G), dim3(kSgThreads), a.L.total_bytes, s, a);
    else if (which == 2) hipLaunchKernelGGL((k_small_grow<int32_t, true>), dim3(io.G), dim3(kSgThreads), a.L.total_bytes, s, a);
    else hipLaunchKernelGGL((k_small_grow<long long, true>), dim3(io.G), dim3(kSgThreads), a.L.total_bytes, s, a);
    return hipGetLastError() == hipSuccess;
}

}  // namespace kern
}  // namespace gbrl
