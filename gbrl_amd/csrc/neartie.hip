// neartie.hip -- near-tie replay: when the exact arg-max of a level has a runner-up within a relative 2^-20 (or a zero gain decides between
// "split" and "leaf"), the few candidates in that window are scored ONCE MORE the way the reference scores them -- float32, its rows in
// ascending order, its operation sequence -- and the winner of THAT comparison is taken, so the tree has the reference's structure even
// where the reference's own rounding noise decided (SURVEY hard part 1; fitter.cpp:318-357 greedy, :411-459 oblivious).
//
// What "the reference's sequence" is (node.cpp:187-251 Cosine, :321-376 L2; math_ops.h:432-449, 476-485, 538-575), as compiled by GCC -O3
// for an AVX2 + FMA target (the reference's release flags; oracle/_ref) -- pinned bit for bit by tests/test_oracle.py against the reference's
// own functions on 6000 random nodes:
//   * per side and output column a float32 sum over the node's rows in ascending row order; mean = sum * (1.0f / count);
//   * sum_{row, col} g[row][col] * mean[col] and sum_col mean[col]^2 are IN-ORDER float32 reductions (`omp simd` without a reduction
//     clause: GCC keeps the order) whose vectorised part rounds the product before the add and whose scalar remainder -- the last
//     n_cols % 4 columns -- is contracted to a fused multiply-add;
//   * Cosine: (num_right + num_left) / sqrtf(fma(norm_right, n_right, norm_left * n_left)), 0 when the denominator is 0;
//     L2: fma(n_left, norm_left, n_right * norm_right);
//   * parent score (greedy, split_candidate_generator.cpp:262-320): the same reductions over all rows of the node; Cosine divides in
//     double: (float)((double)dot / sqrt((double)(norm * n))); L2: norm * n;
//   * gain = fma(score, feature weight, -parent) (greedy; root: parent = 0); oblivious: (sum over the level's nodes in node order) * weight.
// L2 scores are taken on the standardised gradients (g - mean) / (std + 1e-8f), recomputed here from the step's statistics.
//
// The replay is rare (1-3 % of random trees have such a node) and serial by nature -- a float32 sum in a fixed order is a dependent chain --
// so it is written for clarity, not speed: one block per (node, candidate); the two sides' chains run in two waves.
#include "kernels.h"
#include "kernels_common.h"
#include "score_common.h"
#include "neartie_core.h"

#include <algorithm>
#include <vector>

#pragma clang fp contract(off)

namespace gbrl {
namespace kern {
namespace {

constexpr int kNearThreads = 256;


__device__ __forceinline__ float near_gain(const NearTieIO &a, int node, int j) {
    const float sc = a.scores[static_cast<size_t>(node) * a.n_cand + j];
    if (a.oblivious) return sc;   // (per-node term; the level sum is formed by the caller)
    const float par = a.is_root[node] ? 0.0f : a.parent[node];
    return fmaf(sc, a.cand_w[j], -par);
}
// the exact level score of candidate j the way k_argmax_stage1 forms it (fp32 sum in node order, then * w)
__device__ __forceinline__ float near_level_score(const NearTieIO &a, int j) {
    float sc = 0.0f;
    for (int nd = 0; nd < a.n_act; ++nd) sc += a.scores[static_cast<size_t>(nd) * a.n_cand + j];
    return sc * a.cand_w[j];
}

// The candidate CLASSES inside the window, best first, each represented by its lowest reference index.  A class = (gain, rows going right):
// candidates with the same exact gain and the same child sizes induce the same partition of the node -- thresholds between the same two
// rows -- and the reference scores them identically; the same gain with other child sizes is another partition (score_common.h second_merge).
__global__ __launch_bounds__(kNearThreads) void k_near_list(NearTieIO a) {
    const int node = blockIdx.x;
    bool skip = a.near[node] == 0;
    if (!skip && a.max_node_rows > 0) {   // (uniform per block) a node -- or, oblivious, any node of the level -- above the limit keeps the exact arg-max
        if (a.oblivious) { for (int nd = 0; nd < a.n_act; ++nd) skip = skip || a.n_rows[nd] > a.max_node_rows; }
        else skip = a.n_rows[node] > a.max_node_rows;
    }
    if (skip) { if (threadIdx.x == 0) a.list_n[node] = 0; return; }
    __shared__ float sv[kNearThreads];
    __shared__ int si[kNearThreads], sj[kNearThreads], sn[kNearThreads];
    const float b1 = a.best_score[node];
    float mag;
    if (a.oblivious) mag = fabsf(b1);
    else { const float par = a.is_root[node] ? 0.0f : a.parent[node]; mag = fmaxf(fabsf(b1 + par), fabsf(par)); }
    const float lo = b1 - near_window_rel(a.rel, a.oblivious ? a.N : a.n_rows[node]) * mag;
    float prev = INFINITY;
    int prev_n = 0x7fffffff;
    int count = 0;
    // oblivious: every candidate's level score once (n_act loads each), kept in the row-list scratch, which nothing uses yet
    float *lvl = reinterpret_cast<float *>(a.ent);
    if (a.oblivious && !a.lvl_ready) {
        for (int j = threadIdx.x; j < a.n_cand; j += kNearThreads) lvl[j] = near_level_score(a, j);
        __syncthreads();
    }
    for (; count < kNearCands; ++count) {
        float bv = -INFINITY; int bi = 0x7fffffff, bj = -1, bn = 0x7fffffff;
        for (int j = threadIdx.x; j < a.n_cand; j += kNearThreads) {
            const float g = a.oblivious ? lvl[j] : near_gain(a, node, j);
            const int n = a.cand_nr ? a.cand_nr[static_cast<size_t>(node) * a.n_cand + j] : 0;
            if (!(g >= lo) || !(g < prev || (g == prev && n > prev_n))) continue;
            const int r = a.cand_ref[j];
            if (g > bv || (g == bv && (n < bn || (n == bn && r < bi)))) { bv = g; bi = r; bj = j; bn = n; }
        }
        sv[threadIdx.x] = bv; si[threadIdx.x] = bi; sj[threadIdx.x] = bj; sn[threadIdx.x] = bn;
        __syncthreads();
        for (int o = kNearThreads / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) {
                const float ov = sv[threadIdx.x + o]; const int oi = si[threadIdx.x + o], on = sn[threadIdx.x + o];
                const float mv = sv[threadIdx.x]; const int mi = si[threadIdx.x], mn = sn[threadIdx.x];
                if (ov > mv || (ov == mv && (on < mn || (on == mn && oi < mi)))) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; sj[threadIdx.x] = sj[threadIdx.x + o]; sn[threadIdx.x] = on; }
            }
            __syncthreads();
        }
        const float got = sv[0]; const int got_j = sj[0], got_n = sn[0];
        __syncthreads();
        if (got_j < 0) break;
        if (threadIdx.x == 0) a.list[static_cast<size_t>(node) * kNearCands + count] = got_j;
        prev = got;
        prev_n = got_n;
    }
    if (threadIdx.x == 0) a.list_n[node] = count;
}

// oblivious levels of big batches: the level scores with the whole GPU (k_near_list's one block spent 1.3 ms on 32 nodes x 32 768 candidates)
__global__ __launch_bounds__(256) void k_near_level_scores(NearTieIO a) {
    if (a.near[0] == 0) return;
    float *lvl = reinterpret_cast<float *>(a.ent);
    for (int j = blockIdx.x * 256 + threadIdx.x; j < a.n_cand; j += gridDim.x * 256) lvl[j] = near_level_score(a, j);
}

// grid (kNearCands + 1, n_act): block (i, node) replays the node's i-th listed candidate; block (kNearCands, node) the parent score.
// Two passes over the node's rows in ascending order, a tile of kNearTile floats at a time, staged in LDS by all threads (one memory round
// trip per tile): pass 1 the per-side column sums (thread c adds column c's entries in order), pass 2 -- Cosine -- the two in-order dot
// chains, one lane each (waves 0 and 1), over products the staging has already rounded.
__global__ __launch_bounds__(kNearThreads) void k_near_replay(NearTieIO a) {
    const int node = blockIdx.y, i = blockIdx.x;
    const int lnode = a.oblivious ? 0 : node;
    if (a.near[lnode] == 0) return;
    const bool is_parent = i == kNearCands;
    if (is_parent ? (a.oblivious || a.is_root[node]) : (i >= a.list_n[lnode])) return;
    extern __shared__ uint32_t lds[];
    uint32_t *in_map = lds;                         // [2048] bit per row: in the node
    uint32_t *right_map = lds + 2048;               // [2048] bit per row: goes right
    __shared__ int s_cnt[kNearThreads];
    const int n = a.n_rows[node], seg = a.seg_start[node];
    float *out = a.rep + static_cast<size_t>(node) * (kNearCands + 1) + i;
    // the candidate's test (what k_partition applies)
    int fs = 0, bin = -1, is_cat = 0;
    if (!is_parent) {
        const int j = a.list[static_cast<size_t>(lnode) * kNearCands + i];
        fs = a.cand_slot[j];
        const FeatureSlot sl = a.slots[fs];
        is_cat = sl.is_cat;
        bin = is_cat ? (j - sl.cand_base + 1) : (j - sl.cand_base);
    }
    const uint16_t *cbase = a.codes + (static_cast<size_t>(fs >> 4) * a.N) * 16 + (fs & 15);
    if (a.N > kNearMaxRows) {
        // Batches above 65 536 rows (round 6): the bit per row lives in global memory, ONE map per node (k_near_rowmaps), shared by the node's
        // candidate blocks; the side of a row is looked up when the ordered list is written.  Same list, same core, any batch size.
        const uint32_t *gmap = a.maps + static_cast<size_t>(node) * ((static_cast<size_t>(a.N) + 31) >> 5);
        int32_t *ent = a.ent + static_cast<size_t>(seg) * (kNearCands + 1) + static_cast<size_t>(i) * n;
        const int n_words = (a.N + 31) >> 5, per = (n_words + kNearThreads - 1) / kNearThreads;
        const int w0 = min(n_words, static_cast<int>(threadIdx.x) * per), w1 = min(n_words, w0 + per);
        int cnt = 0;
        for (int w = w0; w < w1; ++w) cnt += __popc(gmap[w]);
        s_cnt[threadIdx.x] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int t = 0; t < kNearThreads; ++t) { const int c = s_cnt[t]; s_cnt[t] = run; run += c; }
#ifdef GBRL_NEAR_DEBUG
            if (run != n) printf("[near debug] node %d cand %d: bitmap holds %d rows, the node %d (seg %d, N %d)\n", node, i, run, n, seg, a.N);
#endif
        }
        __syncthreads();
        {
            int pos = s_cnt[threadIdx.x];
            for (int w = w0; w < w1; ++w) {
                uint32_t m = gmap[w];
                while (m) { const int b = __ffs(m) - 1; m &= m - 1; ent[pos++] = (w << 5) + b; }
            }
        }
        __syncthreads();
        // the side of every listed row (coalesced over the list, gathered class codes), and how many go right
        int cr = 0;
        if (!is_parent) {
            for (int p0 = threadIdx.x; p0 < n; p0 += kNearThreads * 8) {     // eight list entries per thread in flight (two dependent round trips each)
                int row[8], code[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) row[u] = ent[min(p0 + u * kNearThreads, n - 1)] & 0x7fffffff;   // (the clamped lanes may see ent[n - 1] AFTER its owner has set the side bit: without the mask their class-code gather goes to a wild address)
#pragma unroll
                for (int u = 0; u < 8; ++u) code[u] = cbase[static_cast<size_t>(row[u]) * 16];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + u * kNearThreads;
                    const bool right = is_cat ? (code[u] == bin) : (code[u] > bin);
                    if (p < n && right) { ent[p] = row[u] | static_cast<int32_t>(0x80000000u); ++cr; }
                }
            }
        }
        __syncthreads();
        s_cnt[threadIdx.x] = cr;
        __syncthreads();
        for (int o = kNearThreads / 2; o > 0; o >>= 1) { if (static_cast<int>(threadIdx.x) < o) s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o]; __syncthreads(); }
        const int n_r = s_cnt[0], n_l = n - n_r;
        __syncthreads();
        if (!is_parent && (n_l < a.min_data || n_r < a.min_data)) { if (threadIdx.x == 0) *out = -INFINITY; return; }   // node.cpp:354
        const NearGrads ng{a.grads, a.meanden, a.D};
        const float res = near_replay_core(ent, n, n_r, ng, a.cosine != 0, is_parent, reinterpret_cast<uint32_t *>(lds), kNearBigTile, kNearBigTileRows);
        if (threadIdx.x == 0) *out = res;
        return;
    }
    for (int w = threadIdx.x; w < 4096; w += kNearThreads) lds[w] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += kNearThreads) {
        const int row = a.rows[seg + p];
        atomicOr(&in_map[row >> 5], 1u << (row & 31));
        if (!is_parent) {
            const int code = cbase[static_cast<size_t>(row) * 16];
            if (is_cat ? (code == bin) : (code > bin)) atomicOr(&right_map[row >> 5], 1u << (row & 31));
        }
    }
    __syncthreads();
    // the node's rows in ascending order, bit 31 = goes right (the partition above leaves a node's segment in no particular order)
    int32_t *ent = a.ent + static_cast<size_t>(seg) * (kNearCands + 1) + static_cast<size_t>(i) * n;
    const int n_words = (a.N + 31) >> 5, per = (n_words + kNearThreads - 1) / kNearThreads;
    const int w0 = min(n_words, static_cast<int>(threadIdx.x) * per), w1 = min(n_words, w0 + per);
    int cnt = 0, cr = 0;
    for (int w = w0; w < w1; ++w) { cnt += __popc(in_map[w]); cr += __popc(in_map[w] & right_map[w]); }
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < kNearThreads; ++t) { const int c = s_cnt[t]; s_cnt[t] = run; run += c; }
    }
    __syncthreads();
    {
        int pos = s_cnt[threadIdx.x];
        for (int w = w0; w < w1; ++w) {
            uint32_t m = in_map[w];
            const uint32_t rm = right_map[w];
            while (m) {
                const int b = __ffs(m) - 1;
                m &= m - 1;
                ent[pos++] = ((w << 5) + b) | static_cast<int32_t>(((rm >> b) & 1u) << 31);
            }
        }
    }
    // rows going right: a block-wide sum of the per-thread counts
    __syncthreads();
    s_cnt[threadIdx.x] = cr;
    __syncthreads();
    for (int o = kNearThreads / 2; o > 0; o >>= 1) { if (static_cast<int>(threadIdx.x) < o) s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o]; __syncthreads(); }
    const int n_r = s_cnt[0], n_l = n - n_r;
    __syncthreads();
    if (!is_parent && (n_l < a.min_data || n_r < a.min_data)) { if (threadIdx.x == 0) *out = -INFINITY; return; }   // node.cpp:354
    const NearGrads ng{a.grads, a.meanden, a.D};
    const float res = near_replay_core(ent, n, n_r, ng, a.cosine != 0, is_parent, reinterpret_cast<uint32_t *>(lds + 4096));
    if (threadIdx.x == 0) *out = res;
}

// N > kNearMaxRows: a bit per row of the batch for every node that will be replayed (list_n > 0; oblivious: every node of a listed level)
__global__ __launch_bounds__(kNearThreads) void k_near_rowmaps(NearTieIO a) {
    const int node = blockIdx.y;
    const int lnode = a.oblivious ? 0 : node;
    if (a.near[lnode] == 0 || a.list_n[lnode] <= 0) return;
    const size_t n_words = (static_cast<size_t>(a.N) + 31) >> 5;
    uint32_t *gmap = a.maps + static_cast<size_t>(node) * n_words;
    const int n = a.n_rows[node], seg = a.seg_start[node];
    // phase 0 (gridDim.x blocks share the words), then phase 1 in a second launch sets the bits
    for (size_t w = static_cast<size_t>(blockIdx.x) * kNearThreads + threadIdx.x; w < n_words; w += static_cast<size_t>(gridDim.x) * kNearThreads) gmap[w] = 0;
    (void)n; (void)seg;
}
__global__ __launch_bounds__(kNearThreads) void k_near_rowmaps_set(NearTieIO a) {
    const int node = blockIdx.y;
    const int lnode = a.oblivious ? 0 : node;
    if (a.near[lnode] == 0 || a.list_n[lnode] <= 0) return;
    uint32_t *gmap = a.maps + static_cast<size_t>(node) * ((static_cast<size_t>(a.N) + 31) >> 5);
    const int n = a.n_rows[node], seg = a.seg_start[node];
    for (int p = blockIdx.x * kNearThreads + threadIdx.x; p < n; p += gridDim.x * kNearThreads) {
        const int row = a.rows[seg + p];
        atomicOr(&gmap[row >> 5], 1u << (row & 31));
    }
}

// ---- round 6: the ORDER of a big node's rows on the whole GPU (the one-lane core's ordering stage redid the bitmap scan in each of a node's 17
// candidate blocks and ranked the places with a pair of barriers per 1024 entries: 1.5 ms per level) ------------------------------------------
constexpr int kNearTileEnt = 2048;
__device__ __forceinline__ bool near_block_active(const NearTieIO &a, int node, int i) {
    const int lnode = a.oblivious ? 0 : node;
    if (a.near[lnode] == 0 || a.list_n[lnode] <= 0) return false;
    return i == kNearCands ? !(a.oblivious || a.is_root[node]) : i < a.list_n[lnode];
}
__device__ __forceinline__ size_t near_tile0(const NearTieIO &a, int node, int i) {    // first tile of block (node, i) in a.tiles; the blocks' tile ranges do not overlap
    return (static_cast<size_t>(a.seg_start[node]) * (kNearCands + 1) + static_cast<size_t>(i) * a.n_rows[node]) / kNearTileEnt + static_cast<size_t>(node) * (kNearCands + 1) + i;
}
// grid (kNearRowRanges, n_act), two launches: the node's bitmap -> its rows in ascending order (once, for all its candidates).  Pass 0 counts the
// set bits of every range of words into a.tiles (free until k_near_sides), pass 1 emits behind the ranges in front (one block per node took
// 0.6 ms for a 2^20-row root).
constexpr int kNearRowRanges = 64;
__global__ __launch_bounds__(256) void k_near_rows(NearTieIO a, int pass) {
    const int node = blockIdx.y, r = blockIdx.x;
    const int lnode = a.oblivious ? 0 : node;
    if (a.near[lnode] == 0 || a.list_n[lnode] <= 0) return;
    __shared__ int s_c[256];
    const uint32_t *gmap = a.maps + static_cast<size_t>(node) * ((static_cast<size_t>(a.N) + 31) >> 5);
    const int n_words = (a.N + 31) >> 5, per_r = (n_words + kNearRowRanges - 1) / kNearRowRanges;
    const int r0 = min(n_words, r * per_r), r1 = min(n_words, r0 + per_r);
    const int per = (r1 - r0 + 255) / 256;
    const int w0 = min(r1, r0 + static_cast<int>(threadIdx.x) * per), w1 = min(r1, w0 + per);
    int cnt = 0;
    for (int w = w0; w < w1; ++w) cnt += __popc(gmap[w]);
    s_c[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) { const int v = static_cast<int>(threadIdx.x) >= o ? s_c[threadIdx.x - o] : 0; __syncthreads(); s_c[threadIdx.x] += v; __syncthreads(); }
    int32_t *cnts = a.tiles + static_cast<size_t>(node) * kNearRowRanges;
    if (pass == 0) { if (threadIdx.x == 255) cnts[r] = s_c[255]; return; }
    int pos = s_c[threadIdx.x] - cnt;
    for (int q = 0; q < r; ++q) pos += cnts[q];
    int32_t *out = a.rowsort + a.seg_start[node];
    for (int w = w0; w < w1; ++w) {
        uint32_t m = gmap[w];
        while (m) { const int b = __ffs(m) - 1; m &= m - 1; out[pos++] = (w << 5) + b; }
    }
}
// grid (tiles, kNearCands + 1, n_act): the side of every listed row, its rank among the tile's right rows (parked in a.pos), the tile's count
__global__ __launch_bounds__(256) void k_near_sides(NearTieIO a) {
    const int node = blockIdx.z, i = blockIdx.y;
    if (!near_block_active(a, node, i)) return;
    const bool is_parent = i == kNearCands;
    const int n = a.n_rows[node], seg = a.seg_start[node];
    int fs = 0, bin = -1, is_cat = 0;
    if (!is_parent) {
        const int j = a.list[static_cast<size_t>(a.oblivious ? 0 : node) * kNearCands + i];
        fs = a.cand_slot[j];
        const FeatureSlot sl = a.slots[fs];
        is_cat = sl.is_cat;
        bin = is_cat ? (j - sl.cand_base + 1) : (j - sl.cand_base);
    }
    const uint16_t *cbase = a.codes + (static_cast<size_t>(fs >> 4) * a.N) * 16 + (fs & 15);
    const size_t off = static_cast<size_t>(seg) * (kNearCands + 1) + static_cast<size_t>(i) * n;
    int32_t *ent = a.ent + off, *pos = a.pos + off;
    const int32_t *rows = a.rowsort + seg;
    const size_t t0 = near_tile0(a, node, i);
    __shared__ int s_c[8 * 4];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    for (int t = blockIdx.x; t * kNearTileEnt < n; t += gridDim.x) {
        int row[8], code[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) row[u] = rows[min(t * kNearTileEnt + u * 256 + static_cast<int>(threadIdx.x), n - 1)];
        if (!is_parent) {
#pragma unroll
            for (int u = 0; u < 8; ++u) code[u] = cbase[static_cast<size_t>(row[u]) * 16];
        }
        unsigned long long mr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = t * kNearTileEnt + u * 256 + static_cast<int>(threadIdx.x);
            const bool right = !is_parent && p < n && (is_cat ? (code[u] == bin) : (code[u] > bin));
            mr[u] = __ballot(right);
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s_c[u * 4 + wv] = __popcll(mr[u]);
        }
        __syncthreads();
        int run = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int wbase = 0, tot = 0;
            for (int w = 0; w < 4; ++w) { const int c = s_c[u * 4 + w]; if (w < wv) wbase += c; tot += c; }
            const int p = t * kNearTileEnt + u * 256 + static_cast<int>(threadIdx.x);
            const bool right = (mr[u] >> lane) & 1ull;
            if (p < n) {
                ent[p] = row[u] | (right ? static_cast<int32_t>(0x80000000u) : 0);
                pos[p] = run + wbase + __popcll(mr[u] & (lane == 0 ? 0ull : (~0ull >> (kWave - lane))));      // right rows of the tile in front of p
            }
            run += tot;
        }
        if (threadIdx.x == 0) a.tiles[t0 + t] = run;
    }
}
// grid (kNearCands + 1, n_act), one wave: exclusive prefix over the block's tiles; the total is the block's number of right rows
__global__ __launch_bounds__(64) void k_near_tilescan(NearTieIO a) {
    const int node = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
    int32_t *nr = a.nr + static_cast<size_t>(node) * (kNearCands + 1) + i;
    if (!near_block_active(a, node, i)) { if (lane == 0) *nr = -1; return; }
    const int n = a.n_rows[node], n_tiles = (n + kNearTileEnt - 1) / kNearTileEnt;
    int32_t *tl = a.tiles + near_tile0(a, node, i);
    int carry = 0;
    for (int b0 = 0; b0 < n_tiles; b0 += kWave) {
        const int b = b0 + lane;
        const int c = b < n_tiles ? tl[b] : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(inc, o, kWave); if (lane >= o) inc += up; }
        if (b < n_tiles) tl[b] = carry + inc - c;
        carry += __shfl(inc, kWave - 1, kWave);
    }
    if (lane == 0) *nr = carry;
}
// grid (tiles, kNearCands + 1, n_act): a listed row's place among the rows of its side
__global__ __launch_bounds__(256) void k_near_places(NearTieIO a) {
    const int node = blockIdx.z, i = blockIdx.y;
    if (!near_block_active(a, node, i)) return;
    const int n = a.n_rows[node];
    const size_t off = static_cast<size_t>(a.seg_start[node]) * (kNearCands + 1) + static_cast<size_t>(i) * n;
    const int32_t *ent = a.ent + off;
    int32_t *pos = a.pos + off;
    const int32_t *tl = a.tiles + near_tile0(a, node, i);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const int before_r = tl[p / kNearTileEnt] + pos[p];
        pos[p] = (static_cast<uint32_t>(ent[p]) >> 31) ? before_r : p - before_r;
    }
}

// ---- round 6: the chains of a big node on the whole GPU (seqsum.hip) ---------------------------------------------------------------------
// (node, i) = block of the replay; its n rows own a region of n * D floats in a.vals: [right: D columns of n_r | left: D columns of n_l], and for
// the dot chains afterwards [right: n_r rows of D products | left: n_l rows].  Chain index = ((node * 17 + i) * 2 + side) * D + c.
// one block: the chain table of a pass (pass 1: one chain per (block, side, column); pass 2: per (block, side)), blocks numbered by a running prefix
__global__ __launch_bounds__(1024) void k_near_chains(NearTieIO a, int pass) {
    SeqChain *ch = static_cast<SeqChain *>(a.chains);
    const int D = a.D, per = pass == 1 ? 2 * D : 2;
    const int n_chains = a.n_act * (kNearCands + 1) * per;
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_chains; c0 += 1024) {
        const int ci = c0 + static_cast<int>(threadIdx.x);
        uint32_t len = 0; const float *x = a.vals;
        if (ci < n_chains) {
            const int blk = ci / per, r = ci - blk * per;
            const int node = blk / (kNearCands + 1), i = blk - node * (kNearCands + 1);
            const int n_r = a.nr[blk];
            if (n_r >= 0) {
                const int n = a.n_rows[node], n_l = n - n_r;
                const float *base = a.vals + (static_cast<size_t>(a.seg_start[node]) * (kNearCands + 1) + static_cast<size_t>(i) * n) * D;
                if (pass == 1) { const int side = r / D, c = r - side * D; len = side ? n_l : n_r; x = base + (side ? static_cast<size_t>(D) * n_r : 0) + static_cast<size_t>(c) * len; }
                else { const int side = r; len = static_cast<uint32_t>(side ? n_l : n_r) * D; x = base + (side ? static_cast<size_t>(D) * n_r : 0); }
            }
        }
        const uint32_t nb = (len + 255u) / 256u;
        s_part[threadIdx.x] = nb;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) { const uint32_t v = threadIdx.x >= static_cast<unsigned>(o) ? s_part[threadIdx.x - o] : 0u; __syncthreads(); s_part[threadIdx.x] += v; __syncthreads(); }
        const uint32_t incl = s_part[threadIdx.x], base = s_base;
        if (ci < n_chains) { ch[ci].x = x; ch[ci].len = len; ch[ci].blk0 = base + incl - nb; ch[ci].start = 0.0f; }
        __syncthreads();
        if (threadIdx.x == 1023) s_base = base + incl;
        __syncthreads();
    }
}
// pass 1 elements: grid (tiles, kNearCands + 1, n_act); a thread takes float4 pieces of listed rows
__global__ __launch_bounds__(256) void k_near_fill1(NearTieIO a) {
#pragma clang fp contract(off)
    const int node = blockIdx.z, i = blockIdx.y;
    const int n_r = a.nr[static_cast<size_t>(node) * (kNearCands + 1) + i];
    if (n_r < 0) return;
    const int n = a.n_rows[node], D = a.D, Q = D >> 2, n_l = n - n_r;
    const size_t off = static_cast<size_t>(a.seg_start[node]) * (kNearCands + 1) + static_cast<size_t>(i) * n;
    const int32_t *ent = a.ent + off, *pos = a.pos + off;
    float *base = a.vals + off * D;
    typedef float near_f4 __attribute__((ext_vector_type(4)));
    for (long long u = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; u < static_cast<long long>(n) * Q; u += static_cast<long long>(gridDim.x) * 256) {
        const int p = static_cast<int>(u / Q), q4 = static_cast<int>(u - static_cast<long long>(p) * Q) * 4;
        const int32_t e = ent[p];
        const int row = e & 0x7fffffff, k = pos[p];
        const bool right = (static_cast<uint32_t>(e) >> 31) != 0;
        const near_f4 v = *reinterpret_cast<const near_f4 *>(a.grads + static_cast<size_t>(row) * D + q4);
        const float g4[4] = {v.x, v.y, v.z, v.w};
        float *col = base + (right ? 0 : static_cast<size_t>(D) * n_r);
        const int len = right ? n_r : n_l;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cc = q4 + t;
            col[static_cast<size_t>(cc) * len + k] = a.meanden == nullptr ? g4[t] : (g4[t] - a.meanden[cc]) / a.meanden[D + cc];   // near_grad's operations
        }
    }
}
__global__ __launch_bounds__(256) void k_near_means(NearTieIO a) {
#pragma clang fp contract(off)
    const int D = a.D, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.n_act * (kNearCands + 1) * 2 * D) return;
    const int blk = t / (2 * D), r = t - blk * 2 * D, side = r / D;
    const int n_r = a.nr[blk];
    if (n_r < 0) return;
    const int node = blk / (kNearCands + 1);
    const int n_l = a.n_rows[node] - n_r;
    const float rec = side ? (n_l > 0 ? 1.0f / static_cast<float>(n_l) : 0.0f) : (n_r > 0 ? 1.0f / static_cast<float>(n_r) : 0.0f);
    a.means[t] = a.sums[t] * rec;
}
// pass 2 elements (Cosine): the rounded products g * mean_side[col] in (row, column) order
__global__ __launch_bounds__(256) void k_near_fill2(NearTieIO a) {
#pragma clang fp contract(off)
    const int node = blockIdx.z, i = blockIdx.y;
    const int blk = node * (kNearCands + 1) + i;
    const int n_r = a.nr[blk];
    if (n_r < 0) return;
    const int n = a.n_rows[node], D = a.D, Q = D >> 2;
    const size_t off = static_cast<size_t>(a.seg_start[node]) * (kNearCands + 1) + static_cast<size_t>(i) * n;
    const int32_t *ent = a.ent + off, *pos = a.pos + off;
    float *base = a.vals + off * D;
    const float *mean = a.means + static_cast<size_t>(blk) * 2 * D;
    typedef float near_f4 __attribute__((ext_vector_type(4)));
    for (long long u = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; u < static_cast<long long>(n) * Q; u += static_cast<long long>(gridDim.x) * 256) {
        const int p = static_cast<int>(u / Q), q4 = static_cast<int>(u - static_cast<long long>(p) * Q) * 4;
        const int32_t e = ent[p];
        const int row = e & 0x7fffffff, k = pos[p];
        const bool right = (static_cast<uint32_t>(e) >> 31) != 0;
        const near_f4 v = *reinterpret_cast<const near_f4 *>(a.grads + static_cast<size_t>(row) * D + q4);
        const float g4[4] = {v.x, v.y, v.z, v.w};
        const float *m = mean + (right ? 0 : D);
        near_f4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cc = q4 + t;
            const float g = a.meanden == nullptr ? g4[t] : (g4[t] - a.meanden[cc]) / a.meanden[D + cc];
            const float pr = g * m[cc];
            if (t == 0) o.x = pr; else if (t == 1) o.y = pr; else if (t == 2) o.z = pr; else o.w = pr;
        }
        *reinterpret_cast<near_f4 *>(base + (right ? 0 : static_cast<size_t>(D) * n_r) + static_cast<size_t>(k) * D + q4) = o;
    }
}
// the scores from the means and the dot sums: the tail of near_replay_core, operation for operation
__global__ __launch_bounds__(64) void k_near_finish(NearTieIO a) {
#pragma clang fp contract(off)
    const int blk = blockIdx.x * 64 + threadIdx.x;
    if (blk >= a.n_act * (kNearCands + 1)) return;
    const int n_r = a.nr[blk];
    if (n_r < 0) return;
    const int node = blk / (kNearCands + 1), i = blk - node * (kNearCands + 1), D = a.D;
    const bool is_parent = i == kNearCands, cosine = a.cosine != 0;
    const int n_l = a.n_rows[node] - n_r;
    float *out = a.rep + blk;
    if (!is_parent && (n_l < a.min_data || n_r < a.min_data)) { *out = -INFINITY; return; }   // node.cpp:354
    const float *mean = a.means + static_cast<size_t>(blk) * 2 * D;
    const float nrf = static_cast<float>(n_r), nlf = static_cast<float>(n_l);
    const float num_r = cosine ? a.sums[static_cast<size_t>(blk) * 2 + 0] : 0.0f, num_l = cosine ? a.sums[static_cast<size_t>(blk) * 2 + 1] : 0.0f;   // (pass 2's chains: [block][side])
    float res;
    if (is_parent) {
        if (cosine) {
            const float den = near_sqnorm(mean + D, D) * nlf;
            res = (n_l == 0 || den == 0.0f) ? 0.0f : static_cast<float>(static_cast<double>(num_l) / sqrt(static_cast<double>(den)));
        } else {
            res = near_sqnorm(mean + D, D) * nlf;
        }
    } else if (cosine) {
        const float tn = near_sqnorm(mean, D), fn = near_sqnorm(mean + D, D);
        const float fden = fn * nlf;
        const float den = fmaf(tn, nrf, fden);
        const float num = num_r + num_l;
        res = den == 0.0f ? 0.0f : num / sqrtf(den);
    } else {
        const float ln = near_sqnorm(mean + D, D), rn = near_sqnorm(mean, D);
        const float rp = nrf * rn;
        res = fmaf(nlf, ln, rp);
    }
    *out = res;
}

// The reference's comparison over the replayed candidates (fitter.cpp:332-357 / 426-459: highest score, first index among equals), written
// where the final arg-max stage reads its input, so that k_resolve_splits (run once more) derives everything else.
__global__ __launch_bounds__(kWave) void k_near_apply(NearTieIO a) {
    const int node = blockIdx.x;
    if (a.near[node] == 0) return;
    const int cnt = a.list_n[node];
    if (cnt <= 0) return;
    if (threadIdx.x == 0) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = 0; i < cnt; ++i) {
            const int j = a.list[static_cast<size_t>(node) * kNearCands + i];
            float g;
            if (a.oblivious) {
                float sc = 0.0f;
                for (int nd = 0; nd < a.n_act; ++nd) sc += a.rep[static_cast<size_t>(nd) * (kNearCands + 1) + i];
                g = sc * a.cand_w[j];
            } else {
                const float par = a.is_root[node] ? 0.0f : a.rep[static_cast<size_t>(node) * (kNearCands + 1) + kNearCands];
                g = fmaf(a.rep[static_cast<size_t>(node) * (kNearCands + 1) + i], a.cand_w[j], -par);
            }
            const int r = a.cand_ref[j];
            if (g > bv || (g == bv && r < bi)) { bv = g; bi = r; }
        }
        a.part_v[static_cast<size_t>(node) * a.n_parts] = bv;
        a.part_i[static_cast<size_t>(node) * a.n_parts] = bi;
    }
    for (int p = 1 + threadIdx.x; p < a.n_parts; p += kWave) {
        a.part_v[static_cast<size_t>(node) * a.n_parts + p] = -INFINITY;
        a.part_i[static_cast<size_t>(node) * a.n_parts + p] = 0x7fffffff;
    }
}

}  // namespace

bool near_tie_supported(int N, int D) { return N >= 1 && N <= (1 << 30) && D >= 1 && D <= kNearMaxD; }
bool near_tie_fast_supported(int N, int D) { return N > kNearMaxRows && D >= 4 && (D & 3) == 0; }
// every listed row contributes D elements to at most one chain per pass, so the chains of a level hold at most 17 N D elements: that many
// 256-element blocks plus one partial block per chain
uint32_t near_tie_fast_blocks(int N, int D, int n_act) {
    const unsigned long long el = static_cast<unsigned long long>(kNearCands + 1) * static_cast<unsigned long long>(N) * D;
    return static_cast<uint32_t>(el / 256 + static_cast<unsigned long long>(n_act) * (kNearCands + 1) * 2 * D + 1);
}
size_t near_tie_fast_chain_bytes(int N, int D, int n_act) {
    return 256 * ((sizeof(SeqChain) * static_cast<size_t>(n_act) * (kNearCands + 1) * 2 * D + 255) / 256) + seq_sums_scratch_bytes(near_tie_fast_blocks(N, D, n_act));
}
size_t near_tie_fast_tiles(int N, int n_act) { return std::max(static_cast<size_t>(kNearCands + 1) * N / kNearTileEnt + static_cast<size_t>(n_act) * (kNearCands + 1) * 2 + 4, static_cast<size_t>(n_act) * kNearRowRanges); }
size_t near_tie_map_words(int N, int n_act) { return N > kNearMaxRows ? ((static_cast<size_t>(N) + 31) >> 5) * static_cast<size_t>(std::max(1, n_act)) : 0; }

void near_tie_replay(const NearTieIO &io, hipStream_t s) {
    const int n_list = io.oblivious ? 1 : io.n_act;
    NearTieIO lio = io;
    if (io.oblivious && io.N > kNearMaxRows && io.n_cand >= 4096) {
        hipLaunchKernelGGL(k_near_level_scores, dim3(std::min(256, (io.n_cand + 255) / 256)), dim3(256), 0, s, io);
        lio.lvl_ready = 1;
    }
    hipLaunchKernelGGL(k_near_list, dim3(n_list), dim3(kNearThreads), 0, s, lio);
    if (io.N > kNearMaxRows) {
        hipLaunchKernelGGL(k_near_rowmaps, dim3(64, io.n_act), dim3(kNearThreads), 0, s, io);
        hipLaunchKernelGGL(k_near_rowmaps_set, dim3(64, io.n_act), dim3(kNearThreads), 0, s, io);
    }
    const bool big = io.N > kNearMaxRows && near_core_words(io.D, kNearBigTile, kNearBigTileRows) * 4 <= 150 * 1024 && 4 * ((io.D + 3) & ~3) <= kNearBigTile;
    const size_t lds = big ? sizeof(uint32_t) * static_cast<size_t>(near_core_words(io.D, kNearBigTile, kNearBigTileRows))
                           : sizeof(uint32_t) * (4096 + static_cast<size_t>(near_core_words(io.D)));
    if (big) {
        static PerDeviceOnce attr;
        if (attr.first() && hipFuncSetAttribute(reinterpret_cast<const void *>(k_near_replay), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) { (void)hipGetLastError(); attr.done = 0; }
    }
    if (!io.fast) hipLaunchKernelGGL(k_near_replay, dim3(kNearCands + 1, io.n_act), dim3(kNearThreads), lds, s, io);
    if (io.fast) {
        const int n_blk = io.n_act * (kNearCands + 1), D = io.D;
        {   // the order: rows once per node, then sides / places of every (node, candidate) list with the whole GPU
            const unsigned tl = static_cast<unsigned>(std::min(512, std::max(1, (io.N + kNearTileEnt - 1) / kNearTileEnt)));
            hipLaunchKernelGGL(k_near_rows, dim3(kNearRowRanges, io.n_act), dim3(256), 0, s, io, 0);
            hipLaunchKernelGGL(k_near_rows, dim3(kNearRowRanges, io.n_act), dim3(256), 0, s, io, 1);
            hipLaunchKernelGGL(k_near_sides, dim3(tl, kNearCands + 1, io.n_act), dim3(256), 0, s, io);
            hipLaunchKernelGGL(k_near_tilescan, dim3(kNearCands + 1, io.n_act), dim3(64), 0, s, io);
            hipLaunchKernelGGL(k_near_places, dim3(tl, kNearCands + 1, io.n_act), dim3(256), 0, s, io);
        }
        SeqChain *chains = static_cast<SeqChain *>(io.chains);
        const size_t table = 256 * ((sizeof(SeqChain) * static_cast<size_t>(n_blk) * 2 * D + 255) / 256);
        void *seq_scratch = static_cast<char *>(io.chains) + table;
        const unsigned tiles = static_cast<unsigned>(std::min<long long>(1024, std::max<long long>(1, (static_cast<long long>(io.N) * (D / 4) + 256 * 16 - 1) / (256 * 16))));
        hipLaunchKernelGGL(k_near_fill1, dim3(tiles, kNearCands + 1, io.n_act), dim3(256), 0, s, io);
        hipLaunchKernelGGL(k_near_chains, dim3(1), dim3(1024), 0, s, io, 1);
        seq_sums(chains, n_blk * 2 * D, io.seq_blocks, seq_scratch, io.sums, nullptr, s);
        hipLaunchKernelGGL(k_near_means, dim3((n_blk * 2 * D + 255) / 256), dim3(256), 0, s, io);
        if (io.cosine) {
            hipLaunchKernelGGL(k_near_fill2, dim3(tiles, kNearCands + 1, io.n_act), dim3(256), 0, s, io);
            hipLaunchKernelGGL(k_near_chains, dim3(1), dim3(1024), 0, s, io, 2);
            seq_sums(chains, n_blk * 2, io.seq_blocks, seq_scratch, io.sums, nullptr, s);
        }
        hipLaunchKernelGGL(k_near_finish, dim3((n_blk + 63) / 64), dim3(64), 0, s, io);
    }
    hipLaunchKernelGGL(k_near_apply, dim3(n_list), dim3(kWave), 0, s, io);
}

bool near_tie_selftest(const float *grads, const uint8_t *in_node, const uint8_t *goes_right, int n_rows, int D, const float *meanden, bool cosine, int min_data, float *out) {
    if (!near_tie_supported(n_rows, D)) return false;
    // the node as the level loop would hand it over: its rows in SCRAMBLED order (the kernel has to restore the ascending order), one
    // numeric feature slot with one threshold, class code 1 = "goes right"
    std::vector<int32_t> rows;
    for (int r = n_rows - 1; r >= 0; --r) if (in_node[r]) rows.push_back(r);
    for (size_t i = 0; i + 2 < rows.size(); i += 3) std::swap(rows[i], rows[i + 2]);
    std::vector<uint16_t> codes(static_cast<size_t>(n_rows) * 16, 0);
    for (int r = 0; r < n_rows; ++r) codes[static_cast<size_t>(r) * 16] = goes_right[r] ? 1 : 0;
    const FeatureSlot slot{0, 1, 0, 0};
    const int32_t zero = 0, one = 1, n_node = static_cast<int32_t>(rows.size());
    const int64_t flag = 1;
    const float w = 1.0f;
    struct Dev { void *p = nullptr; ~Dev() { if (p) (void)hipFree(p); } };
    auto up = [](Dev &d, const void *src, size_t bytes) {
        if (hipMalloc(&d.p, std::max<size_t>(bytes, 16)) != hipSuccess) return false;
        return bytes == 0 || hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    Dev d_rows, d_codes, d_grads, d_md, d_slot, d_zero, d_one, d_n, d_flag, d_w, d_ent, d_rep, d_list;
    bool ok = up(d_rows, rows.data(), rows.size() * 4) && up(d_codes, codes.data(), codes.size() * 2) && up(d_grads, grads, sizeof(float) * n_rows * D) &&
              (!meanden || up(d_md, meanden, sizeof(float) * 2 * D)) && up(d_slot, &slot, sizeof(slot)) && up(d_zero, &zero, 4) && up(d_one, &one, 4) && up(d_n, &n_node, 4) &&
              up(d_flag, &flag, 8) && up(d_w, &w, 4) && up(d_ent, nullptr, 0) && up(d_rep, nullptr, 0) && up(d_list, nullptr, 0);
    if (!ok) { (void)hipGetLastError(); return false; }
    (void)hipFree(d_ent.p); d_ent.p = nullptr;
    (void)hipFree(d_rep.p); d_rep.p = nullptr;
    (void)hipFree(d_list.p); d_list.p = nullptr;
    if (hipMalloc(&d_ent.p, sizeof(int32_t) * static_cast<size_t>(kNearCands + 1) * n_rows) != hipSuccess || hipMalloc(&d_rep.p, sizeof(float) * (kNearCands + 1)) != hipSuccess ||
        hipMalloc(&d_list.p, sizeof(int32_t) * (kNearCands + 1)) != hipSuccess) { (void)hipGetLastError(); return false; }
    NearTieIO io{};
    io.rows = static_cast<int32_t *>(d_rows.p); io.seg_start = static_cast<int32_t *>(d_zero.p); io.n_rows = static_cast<int32_t *>(d_n.p); io.codes = static_cast<uint16_t *>(d_codes.p);
    io.N = n_rows; io.D = D; io.grads = static_cast<float *>(d_grads.p); io.meanden = meanden ? static_cast<float *>(d_md.p) : nullptr; io.cosine = cosine ? 1 : 0; io.oblivious = 0; io.min_data = min_data;
    io.slots = static_cast<FeatureSlot *>(d_slot.p); io.cand_slot = static_cast<int32_t *>(d_zero.p); io.cand_w = static_cast<float *>(d_w.p); io.cand_ref = static_cast<int32_t *>(d_zero.p); io.n_cand = 1;
    io.is_root = static_cast<int32_t *>(d_zero.p); io.near = static_cast<int64_t *>(d_flag.p); io.n_act = 1;
    io.list = static_cast<int32_t *>(d_list.p); io.list_n = static_cast<int32_t *>(d_one.p);   // the one candidate, internal index 0
    io.ent = static_cast<int32_t *>(d_ent.p); io.rep = static_cast<float *>(d_rep.p);
    if (hipMemset(d_list.p, 0, sizeof(int32_t) * (kNearCands + 1)) != hipSuccess) return false;
    const size_t lds = sizeof(uint32_t) * (4096 + static_cast<size_t>(near_core_words(D)));
    hipLaunchKernelGGL(k_near_replay, dim3(kNearCands + 1, 1), dim3(kNearThreads), lds, nullptr, io);
    float rep[kNearCands + 1];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(rep, d_rep.p, sizeof(rep), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
    out[0] = rep[0];
    out[1] = rep[kNearCands];
    return true;
}

}  // namespace kern
}  // namespace gbrl
