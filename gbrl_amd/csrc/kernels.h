// kernels.h -- launch wrappers of the gfx950 kernels behind GBRL::step / GBRL::predict.
// Each wrapper enqueues on `stream` and returns immediately; errors surface through hipGetLastError in the engine.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace gbrl {
namespace kern {

// hipFuncSetAttribute (and anything else that is per-device state) must run once per DEVICE, not once per process: one process
// can hold models on several GPUs.  `static PerDeviceOnce once; if (once.first()) hipFuncSetAttribute(...)`.
struct PerDeviceOnce {
    uint64_t done = 0;
    bool first() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        if ((done >> d) & 1ull) return false;
        done |= 1ull << d;
        return true;
    }
};


// A contiguous run of positions [start, start+len) of the row list that belongs to one slot (node or leaf).
struct Chunk {
    int32_t slot;
    int32_t start;
    int32_t len;
    int32_t pad;
};

// Descriptor of one feature slot (numeric feature or categorical feature) for the scoring kernel.
struct FeatureSlot {
    int32_t is_cat;     // 0 numeric: candidate k <-> threshold k, right = classes > k; 1 categorical: candidate j <-> class j+1
    int32_t n_cand;     // number of candidates of this slot
    int32_t cand_base;  // index of the slot's first candidate in the global candidate order
    int32_t pad;
};

// Per-node split decision used by the partition kernel.
struct NodeSplit {
    int32_t do_split;   // 0: copy the segment unchanged
    int32_t fslot;      // feature slot of the chosen candidate
    int32_t bin;        // numeric: threshold index k (right <=> code > k); categorical: class id (right <=> code == bin)
    int32_t is_cat;
    int32_t seg_start;  // first position of the node's segment
    int32_t n_left;     // rows going left (right child starts at seg_start + n_left)
    uint32_t thr_key;   // numeric: ordered key of the threshold -- right <=> key(row, feature) > thr_key, read from the
                        // feature-major keys (4 coalesced bytes per row instead of a 32-byte code record)
    int32_t pad1;
};

// Fixed-point scales of one step, derived from the gradient statistics.  They live in device memory so that the whole
// statistics -> scale -> quantisation chain runs without a host round trip (the host reads them once, later).
struct StepScales {
    double inv_scale;   // 2^-sbits : histogram sums -> standardised-gradient units (k_score)
    double leaf_scale;  // 2^lbits  : raw-gradient fixed point of the leaf sums (k_leaf_sums)
    float scale;        // 2^sbits  (k_quantize)
    float hmax_build;   // max |standardised gradient| * 1.0001 (or max |g| for Cosine)
    float hmax_raw;     // max |g|
    int32_t sbits, lbits, pad;
};

constexpr int kMaxPath = 32;
constexpr int kPartitionRows = 4096;  // rows per partition block (chunk size the engine must use)  // max_depth supported by the duplicate-on-path check

// ---- gradient preprocessing (A2) ----
// out[0..D) = sum_i g[i,d] (center == null) or sum_i (g[i,d]-center[d])^2 ; out[D..2D) = max_i |g[i,d] (- center[d])|
// block_partials: n_blocks * 2D doubles.  Deterministic (fixed reduction tree).
void column_sums(const float *g, int n, int D, const float *center /*nullable [D]*/, double *block_partials,
                 int n_blocks, double *out /*[2D]*/, hipStream_t s);
int column_sums_blocks(int n, int D);
// max |standardised g| as float bits in out[0]; mean/inv_std nullable (Cosine: raw grads)
// Device-side statistics chain (one GPU): mean from the column sums; then std + 1e-8, the maxima and both fixed-point scales.
void stats_mean(const double *stat /*[2D] sums | max*/, long long n, int D, float *meanden /*[2D] mean | denom*/, hipStream_t s);
void stats_finish(const double *stat_raw /*[2D] sums | max|g|*/, const double *stat_centred /*nullable [2D] sum sq | max|g-mean|*/,
                  long long n, int D, int chunk_rows, float *meanden, StepScales *sc, hipStream_t s);
// RL-sized batches on one GPU: the whole chain above + the quantisation in one launch, same bits (k_small_stats); false: not covered
bool small_stats(const float *g, int n, int D, bool centred, int chunk_rows, double *stat /*[4D]*/, float *meanden /*[2D]*/, StepScales *sc, int32_t *qg,
                 hipStream_t s);
// qg = rint(((g - mean) / denom) * scale) as int32  (mean/denom nullable)
void quantize_grads(const float *g, size_t n_el, int D, const float *mean, const float *denom, const StepScales *sc,
                    int32_t *qg, hipStream_t s);

// ---- split candidates (A3, A4) ----
void column_minmax(const uint32_t *kt /*feature-major keys [F][n]*/, int n, int F, uint32_t *mn, uint32_t *mx, hipStream_t s);  // ordered keys
void uniform_thresholds(const uint32_t *min_keys, const uint32_t *max_keys, int F, int B, float *thr, hipStream_t s,
                        uint32_t *thr_keys = nullptr /*the thresholds' ordered keys, written by the same launch*/);
// counts[f][j] += #{rows : j == #{k : trial[f][k] (strict ? < : <=) key(x)}}; optionally writes codes[row*code_stride + code_off + f]
void bin_rows(const float *obs, int n, int F, const uint32_t *trial_keys, int B, bool strict, int64_t *counts,
              uint16_t *codes, int code_stride, int code_off, hipStream_t s);
void qsel_init(uint32_t *prefix, uint32_t *trial, int F, int B, hipStream_t s);
// one bisection step on bit `bit`: uses counts of trial = prefix|bit; then prepares trial for `next_bit` (or final keys if <0)
void qsel_update(uint32_t *prefix, uint32_t *trial, const int64_t *counts, const int64_t *cum_ranks, int F, int B,
                 int bit, int next_bit, hipStream_t s);
// fit(): MultiRMSE gradients g = pred - target (loss.cpp:42-56) and a row gather for the shuffled copy of the data set
void sub_arrays(const float *a, const float *b, float *out, size_t n, hipStream_t s);
void gather_rows(const float *src, const int32_t *perm, float *dst, int n, int width, hipStream_t s);
void f64_to_f32(const double *in, float *out, int n, hipStream_t s);
// row-sharded statistics: [D sums | P x D maxima, one row per rank] summed over ranks = sums + gathered maxima (kernels.hip)
// (+ one trailing word, `extra`, summed like the rest: this rank's row count in the first round)
void stats_pack(const double *st, int D, int P, int rank, double *msg, hipStream_t s, double extra = 0.0);
void stats_unpack(const double *msg, int D, int P, double *st, hipStream_t s);
void negate_f32(float *p, int n, hipStream_t s);
void f32_to_f64(const float *in, double *out, int n, hipStream_t s);
void keys_to_floats(uint32_t *keys /*NaN-range keys are raised to -inf's key in place*/, float *out, size_t n, hipStream_t s);
void floats_to_keys(const float *in, uint32_t *keys, size_t n, hipStream_t s);
// The level's result block handed to the host without a copy engine: one block copies `bytes` (a multiple of 4) from device memory to
// pinned, device-mapped host memory and then stores `seq` to `flag` (system scope).  The host polls the flag; what it then reads is
// complete (the stores are fenced before the flag) and every earlier operation of the stream has finished.
void publish_block(void *d_src, void *h_dst_mapped, size_t bytes, uint32_t *flag_mapped, uint32_t seq, hipStream_t s, bool zero_src = false /*clear the source words after copying them*/);
// Two device arrays (sizes multiples of 4 bytes) into pinned, device-mapped host memory with one launch, no flag.
void publish_pair(const void *d_a, void *h_a_mapped, size_t a_bytes, const void *d_b, void *h_b_mapped, size_t b_bytes, hipStream_t s);
// Up to 8 small host -> device uploads with ONE launch: the sources live in pinned, device-mapped host memory (pass their DEVICE
// aliases), every segment is a multiple of 4 bytes.  Replaces a handful of copy-engine transfers (each ~10 us of stream time for a
// few KB) in steps that are bound by stream operations, not bytes.  The host must leave the sources alone until the kernel has run.
struct FetchSegments { void *dst[8]; const void *src[8]; uint32_t words[8]; int n; };
void fetch_segments(const FetchSegments &fs, hipStream_t s);
// up to 16 byte ranges out of one mapped pinned staging block by ONE launch (the model's device mirror after a step: a dozen appended slices
// of a few hundred bytes each; a hipMemcpyAsync per slice from pageable memory cost ~10 us apiece).  Ranges need no alignment.
constexpr int kStageSegments = 32;   // (a greedy model appends 17-18 slices per tree: one launch, not two)
struct StageSegments { void *dst[kStageSegments]; uint32_t src_off[kStageSegments]; uint32_t bytes[kStageSegments]; int n; };
void stage_copy(const StageSegments &ss, const void *stage_mapped, hipStream_t s);
// up to four regions filled with a 32-bit pattern each by ONE launch (an RL-sized step pays 4-5 us per hipMemsetAsync)
struct FillSegments { void *dst[4]; uint32_t words[4]; uint32_t value[4]; int n; };
void fill_segments(const FillSegments &fs, hipStream_t s);
void iota_rows(int32_t *rows, int n, hipStream_t s);

// ---- exact quantile selection and binning on transposed keys (quantile.hip) ----
struct QuantilePlan { int sample, n_split, n_chunks, chunk; };
constexpr int kQuantileClasses = 8192;
constexpr int kQuantileMaxSplit = 4095;
QuantilePlan quantile_plan(int n);
void transpose_keys(const float *obs, int n, int F, uint32_t *kt /*[F][n]*/, hipStream_t s);
void sample_splitters(const uint32_t *kt, int n, int F, const QuantilePlan &p, uint32_t *splitters /*[F][4095] sorted*/,
                      uint32_t *splitters_bfs /*[F][4095] breadth-first order, used by the searches*/, hipStream_t s);
void class_count(const uint32_t *kt, int n, int F, const QuantilePlan &p, const uint32_t *splitters_bfs,
                 uint32_t *partial /*[n_chunks][F][8192]*/, hipStream_t s);
// global_counts (nullable): all-reduced class counts of a row-sharded run -- ranks come from them, list lengths stay local
void quantile_targets(const uint32_t *partial, const int64_t *global_counts, const uint32_t *splitters, const int64_t *cum, int F, int B,
                      const QuantilePlan &p, uint32_t *class_off /*[F][8192], preset 0xff*/, uint32_t *tgt_off, uint32_t *tgt_len,
                      uint32_t *tgt_rank, uint32_t *thr_keys, uint32_t *alloc /*[1], zeroed*/, uint32_t max_elems,
                      uint32_t *overflow /*zeroed*/, hipStream_t s);
// row-sharded selection: raw local sample -> exchange -> identical splitters on every rank; bisection on the extracted lists
void sample_only(const uint32_t *kt, int n, int F, int S, uint32_t *sample_out /*[F][S]*/, hipStream_t s);
void place_sample(const uint32_t *samp, int F, int S, int rank, int SU, int64_t *uni /*[F][SU], zeroed*/, hipStream_t s);
void union_splitters(const int64_t *uni, int F, int SU, int n_split, uint32_t *splitters, uint32_t *splitters_bfs, hipStream_t s);
void counts_to_i64(const uint32_t *partial, int n_chunks, size_t fc, int64_t *out, hipStream_t s);
void select_count(const uint32_t *lists, const uint32_t *tgt_off, const uint32_t *tgt_len, const uint32_t *prefix, int bit, int n_targets,
                  int64_t *counts, hipStream_t s);
void select_update(uint32_t *prefix, const int64_t *counts, const uint32_t *tgt_off, const uint32_t *tgt_rank, int bit, int n_targets,
                   uint32_t *thr_keys, hipStream_t s);
void quantile_extract(const uint32_t *kt, int n, int F, const QuantilePlan &p, const uint32_t *splitters_bfs, const uint32_t *class_off,
                      const uint32_t *partial, uint32_t *out, hipStream_t s);
void quantile_select(const uint32_t *lists, const uint32_t *tgt_off, const uint32_t *tgt_len, const uint32_t *tgt_rank, int n_targets,
                     uint32_t *thr_keys, hipStream_t s);
// codes[(slot/16)*n*16 + row*16 + slot%16] (u16) = #{k : thr_key[f][k] < key(row, f)}
// small batches: one LDS sort per column (n <= sort_quantiles_max_rows()); thr_keys[f][k] = key of 1-based rank cum[k]
int sort_quantiles_max_rows();
bool sort_quantiles_fits(int n, int B);   // the LDS sort takes (S + B) * 4 bytes, S = n rounded up to a power of two: <= 64 KiB
void sort_quantiles(const uint32_t *kt, int n, int F, const int64_t *cum, int B, uint32_t *thr_keys, float *thr_floats /*the same thresholds as floats*/, hipStream_t s, uint16_t *codes = nullptr /* also write the class codes (k_bin_cols' output) */);
// radix_select.hip: exact order statistics by MSD radix counting (one GPU)
size_t radix_state_bytes(int F, int B);
size_t radix_partial_bytes(int F);
int radix_max_targets();
size_t radix_list_bytes(int n, int F);
// Row-sharded runs pass the exchange hook and two scratch buffers; cum then holds GLOBAL ranks and every rank ends with the
// same thresholds.  Returns 0, or non-zero when a HIP call (1) / the all-reduce (2) failed.
struct RadixComm {
    void *ctx;
    int (*allreduce_sum_i64)(void *ctx, int64_t *dev_buf, size_t count);   // called with the stream synchronised unless stream_ordered
    bool stream_ordered;        // the all-reduce is enqueued on the stream (native RCCL): radix_select never waits for the device -- the messages are
                                // sized by their upper bound (B slot rows per feature) instead of the slot counts read back after every pass
    int64_t *gbuf;              // radix_exchange_words(F) int64
    uint32_t *partial_global;   // radix_global_partial_bytes(F)
};
size_t radix_exchange_words(int F);
size_t radix_global_partial_bytes(int F);
int radix_select(const uint32_t *kt, int n, int F, const int64_t *cum, int B, void *state, uint32_t *partial, uint32_t *lists,
                 uint32_t *thr_keys, hipStream_t s, const RadixComm *comm /*nullable: one GPU*/,
                 int pass1_chunks = 0 /*> 0: `partial` already holds the first-digit counts written by transpose_keys_count*/,
                 uint32_t *le_out = nullptr /*[F][B] + [F]: number of keys <= each threshold (global, after the NaN-range fix-up of
                 keys_to_floats), for the root's class counts; the F trailing words are scratch*/);
// obs [n][F] -> kt [F][n] keys AND the first radix digit counted on the way (partial: the radix_partial_bytes(F) buffer).
// Returns the chunk count to pass to radix_select, or 0 when the shape does not qualify (nothing was done).
int transpose_keys_count(const float *obs, int n, int F, uint32_t *kt, uint32_t *partial, hipStream_t s);
void bin_cols(const uint32_t *kt, int n, int F, const uint32_t *thr_keys, int B, uint16_t *codes, hipStream_t s);
void scatter_cat_codes_grouped(const uint16_t *cat_codes, int n, int Fc, int F, uint16_t *codes, hipStream_t s);
constexpr int kCodeGroup = 16;  // code layout: groups of 16 feature slots, [group][row][16]

// ---- split-score histograms (A6) ----
size_t hist_lds_bytes(int NB, int D, int FG);
// HistDirect (optional): every node of the level is ONE chunk (RL-sized batches), so a block's LDS tile already holds the node's
// complete sums for its feature group and leaves as hist[slot_map ? slot_map[chunk.slot] : chunk.slot][feature][class][D+1] (int64)
// -- no partials, no hist_reduce launch.  hist_build returns true when it wrote the histograms this way (the wide / quad variants
// for more than 16 outputs do not: the caller then reduces the partials as usual).
struct HistDirect {
    int64_t *hist = nullptr;
    const int32_t *slot_map = nullptr;
    int Fp = 0;
};
bool hist_direct_supported(int FG);   // the kernel hist_build will take for this layout can store the histograms itself
bool hist_build(const uint16_t *codes, int n_rows, const int32_t *qg, int D, const int32_t *rows,
                const Chunk *chunks, int n_chunks, int n_groups, int FG, int NB, int32_t *partials, hipStream_t s,
                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr /*the dispatch's own begin / end timestamps*/,
                const HistDirect *direct = nullptr,
                bool count_rows = true /* false: the rows' count field is NOT accumulated (FG == 16, D <= 16 kernels only; returns as if
                true otherwise -- check hist_countless_supported): the root's class counts come from the selection, hist_reduce root_le */);
bool hist_countless_supported(int D, int FG, int n_rows);
// hist[slot_map ? slot_map[k] : k] = sum of the partials of chunk-slot k
void hist_reduce(const int32_t *partials, const int32_t *slot_chunk_begin /*[n_slots+1]*/, const int32_t *slot_map, int n_slots,
                 int n_groups, int FG, int NB, int D, int Fp, int64_t *hist, hipStream_t s,
                 int chunks_per_slot = 1 /* average chunks per node: picks the kernel variant */,
                 int scatter_fs = 0 /* > 0: write the feature-scattered send layout [f / fs][n_slots][f % fs][class][D+1] of a
                                       reduce-scatter over feature slices of fs features instead of hist[slot][Fp][class][D+1] */,
                 const uint32_t *root_le = nullptr /* [F][B] keys <= threshold: the count field of numeric feature f, class c becomes
                                       le[c] - le[c-1] (c = B: n_total - le[B-1]) instead of the partials' sum (root level, FG == 16) */,
                 int root_F = 0, int root_B = 0, long long root_n = 0);
// Row-sharded runs, feature-parallel scoring: the reduce-scattered slice [n][fs][class][D+1] of this rank's features goes to its
// place hist[slot_map[k]][lo + f][class][D+1] (f < min(fs, Fp - lo))
void hist_place_slice(const int64_t *recv, int64_t *hist, const int32_t *slot_map, int n, int fs, int lo, int Fp, size_t feat_elems, hipStream_t s);
void fill_f32(float *p, size_t n, float v, hipStream_t s);
// Winner of a level across ranks with ONE small sum all-reduce: every rank fills its own row of gather[P][n_win + 2 * n_act]
// (order-preserving key of (local best score, lowest reference index) per winner slot; total | right row counts of ITS best
// candidate per node) -- the other rows stay zero, so the sum is an all-gather -- and winner_adopt picks the highest key (ties:
// lowest index, the order of the single-GPU arg-max) and rebuilds best_idx / best_score / counts4 / the NodeSplit descriptors from
// the owning rank's counts.  n_win = 1 (oblivious: one condition per level) or n_act (greedy).
void winner_pack(const int32_t *best_idx, const float *best_score, const int64_t *counts4, int max_front, int n_win, int n_act, int rank,
                 int64_t *gather /*[P][n_win + 2 * n_act]: the kernel clears the other ranks' rows*/, hipStream_t s, int P);
void winner_adopt(const int64_t *gather, int P, int n_win, int n_act, bool oblivious, const int32_t *ref_to_internal, const int32_t *cand_slot,
                  const FeatureSlot *slots, const int32_t *seg_start, const uint32_t *thr_keys, int B, int32_t *best_idx, float *best_score,
                  int64_t *counts4, int max_front, NodeSplit *out, int32_t *cursors, hipStream_t s);
// cur[dst] = prev[parent] - cur[sibling]  (entries: triples {dst, parent, sibling or -1})

// ---- scoring / selection (A6, A7, A8) ----
// sub_par[node] >= 0: the node's histogram is hist_prev[sub_par] - hist[sub_sib] (sub_sib < 0: no sibling rows) -- computed,
// scored AND written back by the kernel (fused sibling subtraction); sub_par nullable
void score_candidates(int64_t *hist, const int64_t *hist_prev, const int32_t *sub_par, const int32_t *sub_sib, int n_nodes, int Fp, int NB, int D, const FeatureSlot *slots, int n_slots,
                      const float *thr /*[F][B]*/, int B, int n_cand, int min_data, int cosine, const StepScales *sc,
                      const int32_t *path_len, const int32_t *path_slot, const float *path_val, const int32_t *path_bin,
                      float *scores /*[n_nodes][n_cand]*/, float *parent /*[n_nodes]*/,
                      const float *cand_w, const int32_t *cand_ref, const int32_t *is_root,
                      float *part_v /*nullable.  greedy: [n_nodes][n_slots] best gain per feature -- arg-max stage 1 fused, scores not written*/,
                      int32_t *part_i, hipStream_t s, int slot0 = 0 /* first feature slot scored (n_slots of them): feature-parallel scoring */,
                      bool keep_derived = true /* false (last level only): derived slices are scored but not written back; resolve_splits
                      must then be given hist_prev / sub_par / sub_sib */,
                      float *part_s = nullptr /* greedy, nullable: [n_nodes][n_slots] the best gain of another candidate class than part_v's (near-tie detection) */,
                      int32_t *part_n = nullptr /* with part_s: rows the slot's best sends right (0 when it sends all of them: one class with 'none') */,
                      int32_t *cand_nr = nullptr /* nullable, scores mode (part_v == nullptr): [n_nodes][n_cand] rows every candidate sends right */);
// best_idx holds REFERENCE candidate indices (cand_ref[j]); ties go to the lowest reference index.  oblivious: one
// result (sum over nodes); greedy: one per node.  part_v/part_i: scratch of n_nodes * argmax_parts(n_cand).
int argmax_parts(int n_cand);
void argmax(const float *scores, int n_nodes, int n_cand, const float *cand_weight, const int32_t *cand_ref, const float *parent,
            const int32_t *is_root, bool oblivious, float *part_v, int32_t *part_i, int32_t *best_idx, float *best_score,
            hipStream_t s, float *part_s = nullptr /* nullable: per block the best score strictly below part_v */);

// near-tie detection inside resolve_splits (one GPU): counts4[2 * max_front + node] = 1 when the best DISTINCT runner-up is within
// rel * (magnitude of the scores) of the winner, or (greedy) the winning gain is that close to zero
struct NearDetect { const float *part_s; const int32_t *part_n /* nullable: oblivious */; float rel; const float *parent; const int32_t *is_root; int cosine; long long rows; /* of the batch (oblivious levels) */ };

// winner -> (feature slot, class) and the child sizes it induces, per active node (one read-back per level)
void resolve_splits(const float *part_v, const int32_t *part_i, int n_parts /*argmax stage-1 output; the final stage runs here*/,
                    int32_t *best_idx, float *best_score, bool oblivious, int n_nodes, const int32_t *ref_to_internal, const int32_t *cand_slot,
                    const FeatureSlot *slots, const int64_t *hist_local, const int64_t *hist_global /*nullable*/, int Fp, int NB, int D,
                    NodeSplit *out, int64_t *counts4 /*[4][max_front]*/, int max_front,
                    const int32_t *seg_start /*nullable: when given, out[] is a complete partition descriptor (do_split from
                    best_score, seg_start, n_left from hist_local) and cursors[2*node..] are zeroed*/,
                    int32_t *cursors, const uint32_t *thr_keys /*[F][B] ordered threshold keys*/, int B, hipStream_t s,
                    void *pub = nullptr /*pinned, device-mapped copy of the result block [best_idx | best_score | counts4]: mirrored, then ...*/,
                    uint32_t *pub_flag = nullptr /*... the last block stores pub_seq here (system scope)*/, uint32_t pub_seq = 0,
                    unsigned *pub_done = nullptr /*device counter, zero between launches*/,
                    const int64_t *hist_prev = nullptr, const int32_t *sub_par = nullptr /*non-null: derived nodes' counts = hist_prev[par] - hist_local[sib]*/,
                    const int32_t *sub_sib = nullptr, const NearDetect *near = nullptr);

// ---- near-tie replay (neartie.hip): the candidates inside the window of a flagged node are re-scored in the reference's float32
// sequence, and the arg-max input of resolve_splits (part_v / part_i) is rewritten with the reference's comparison of them.
constexpr int kNearCands = 16;         // distinct gains replayed per node (the closest ones)
constexpr int kNearMaxRows = 65536;    // batches up to this many rows keep the replay's row bitmaps in LDS; larger ones use a per-node bitmap in global memory (round 6)
constexpr int kNearMaxD = 1024;      // (the replay block keeps two mean vectors, a tile and two row bitmaps in 64 KB of LDS)
struct NearTieIO {
    const int32_t *rows;        // the level's row list
    const int32_t *seg_start;   // [n_act] first position of every active node's segment
    const int32_t *n_rows;      // [n_act] its length
    const uint16_t *codes;      // [group][row][16] class codes
    int N, D;
    const float *grads;         // [N][D] raw gradients
    const float *meanden;       // L2: [D] mean | [D] std + 1e-8f of the step; nullptr: raw gradients (Cosine)
    int cosine, oblivious, min_data;
    const FeatureSlot *slots;
    const int32_t *cand_slot;   // internal candidate -> feature slot
    const float *cand_w;        // [n_cand] feature weight of every internal candidate
    const int32_t *cand_ref;    // [n_cand] reference index of every internal candidate
    int n_cand;
    const float *scores;        // [n_act][n_cand] exact scores (score_candidates without part_v)
    const int32_t *cand_nr;     // greedy: [n_act][n_cand] rows every candidate sends right (same launch); nullptr: oblivious
    const float *parent;        // [n_act] exact parent scores
    const int32_t *is_root;     // [n_act]
    const float *best_score;    // [n_act] exact best gain (oblivious: [0])
    const int64_t *near;        // [n_act] flags (oblivious: [0])
    float rel;
    int n_act;
    int32_t *list;              // scratch [n_act][kNearCands]
    int32_t *list_n;            // scratch [n_act]
    int32_t *ent;               // scratch [max((kNearCands + 1) * N, n_cand)]: ordered row lists (the candidate list of an oblivious level keeps its level scores there first)
    float *rep;                 // scratch [n_act][kNearCands + 1]: replayed scores, [kNearCands] = the parent's
    float *part_v;              // arg-max stage-1 arrays of the level: rewritten for the replayed nodes
    int32_t *part_i;
    int n_parts;
    // N > kNearMaxRows and D % 4 == 0 (round 6): the chains are evaluated by seqsum.hip on the whole GPU instead of one lane per chain
    int lvl_ready;              // oblivious: the level scores of all candidates already sit in `ent` (k_near_level_scores ran: big levels)
    int fast;                   // 1: the order, the chains and the scores are produced by the whole GPU (k_near_rows .. k_near_finish)
    int32_t *pos;               // scratch [(kNearCands + 1) * N]: a listed row's place among the rows of its side
    int32_t *nr;                // scratch [n_act][kNearCands + 1]: rows going right (-1: block not replayed)
    int32_t *rowsort;           // scratch [N]: every replayed node's rows in ascending order (one list per node, shared by its candidates)
    int32_t *tiles;             // scratch [near_tie_fast_tiles]: per 2048-entry tile of a (node, candidate) list: rows going right, then their exclusive prefix
    float *vals;                // scratch [(kNearCands + 1) * N * D]: the chains' elements (column-major per side, then the dot chains' products)
    float *means;               // scratch [n_act][kNearCands + 1][2][D]
    float *sums;                // scratch [n_act][kNearCands + 1][2][D]: chain results (pass 1: per column; pass 2: [side][0])
    void *chains;               // scratch: SeqChain table [n_act * (kNearCands + 1) * 2 * D] + seq_sums scratch behind it
    size_t chains_bytes;
    uint32_t seq_blocks;        // upper bound of the chains' 256-element blocks
    uint32_t *maps;             // N > kNearMaxRows: scratch [n_act][ceil(N / 32)] -- a bit per row of the batch, set for the node's rows (k_near_rowmaps)
    int max_node_rows;          // nodes of more rows are not replayed (an oblivious level: when ANY of its nodes is larger); 0: no limit
};
size_t near_tie_map_words(int N, int n_act);   // 0 when the batch fits the LDS bitmaps
bool near_tie_fast_supported(int N, int D);     // big batch, D a multiple of 4: chains through seqsum.hip
uint32_t near_tie_fast_blocks(int N, int D, int n_act);
size_t near_tie_fast_chain_bytes(int N, int D, int n_act);
size_t near_tie_fast_tiles(int N, int n_act);
bool near_tie_supported(int N, int D);
void near_tie_replay(const NearTieIO &io, hipStream_t s);
// diagnostics (gbrl_hip_replay_scores): one node given by per-row flags, host pointers; out[0] = split score, out[1] = parent score
bool near_tie_selftest(const float *grads, const uint8_t *in_node, const uint8_t *goes_right, int n_rows, int D, const float *meanden, bool cosine, int min_data, float *out);

// seqsum.hip: sequential float32 sums of many arrays ("chains"), evaluated in parallel and bit for bit (block summaries that depend on the
// running sum's parity only; the serial loop where a power of two is crossed)
struct SeqChain { const float *x; uint32_t len; uint32_t blk0; float start; };   // blk0: first of the chain's ceil(len / 256) blocks in the flat block arrays (ascending over the chains)
size_t seq_sums_scratch_bytes(uint32_t n_blocks);
void seq_sums(const SeqChain *d_chains, int n_chains, uint32_t n_blocks, void *d_scratch, float *d_out, uint32_t *d_n_slow /*nullable*/, hipStream_t s);
bool seq_sums_selftest(const float *x, const uint32_t *lens, const float *starts /*nullable*/, int n_chains, float *out, uint32_t *n_slow_out /*nullable*/);

// rows going right per node for the chosen splits (row-sharded runs: local child sizes without a local histogram)
void localize_splits(NodeSplit *splits, const int32_t *n_local, const int64_t *right_local, int n_nodes, hipStream_t s);
void localize_publish(NodeSplit *splits, const int32_t *n_local, const int64_t *right_local, int n_nodes, void *d_src, void *h_dst_mapped, size_t bytes,
                      uint32_t *flag_mapped, uint32_t seq, hipStream_t s);   // localize_splits + publish_block in one launch
void hist_place(const int64_t *src, int64_t *dst, const int32_t *slot_map, int n, size_t node_elems, hipStream_t s);
void count_right(const int32_t *rows, const uint16_t *codes, const uint32_t *kt /*nullable: feature-major keys*/, int n_rows, const Chunk *chunks, int n_chunks, const NodeSplit *splits,
                 int64_t *n_right /*[n_nodes], zeroed*/, hipStream_t s);

// ---- device-side level planning (oblivious trees on one GPU) ----
// All descriptors of tree level L -- node segments, the histogram chunk table of the smaller child of every pair, slot maps for the
// sibling subtraction, path arrays, partition chunks -- built on the device from the previous level's resolved splits, so that a
// whole tree is enqueued without a host round trip per level (the host replays the bookkeeping from the per-level result blocks
// after ONE synchronisation).  Unused entries of the chunk tables (up to cap_h / cap_p) get len 0: the consumers are launched with
// those worst-case grids and empty chunks exit at once.  state[0] != 0: growth has stopped (fitter.cpp:458), state[1] = level.
struct ObliviousPlan {
    int32_t *node_seg, *node_n;            // [max_depth + 1][mf] segment start / row count of the level's nodes
    int mf;
    Chunk *chunks; int cap_h;              // histogram chunks of the level
    int32_t *chunk_begin;                  // [mf + 1]
    int32_t *slot_map, *sub_par, *sub_sib; // [mf]
    int32_t *path_len, *path_slot, *path_bin, *is_root;   // [mf] / [mf][kMaxPath]
    float *path_val;
    Chunk *part_chunks; int cap_p;         // partition chunks of the level
    int32_t *seg_starts;                   // [mf]
    int32_t *state;                        // [2]
    int32_t *cond_slot, *cond_bin;         // [kMaxPath] the conditions chosen so far (one per level)
    float *cond_val;
};
void plan_oblivious_level(int level, int n_rows, int chunk_rows, int budget, const NodeSplit *resolved_prev, const float *best_score_prev,
                          const float *thr, int B, const ObliviousPlan &pl, hipStream_t s);

// ---- partition (A9) ----
void partition_rows(const int32_t *rows_in, int32_t *rows_out, const uint16_t *codes, const uint32_t *kt /*nullable*/, int n_rows,
                    const Chunk *chunks, int n_chunks, const NodeSplit *splits, int32_t *cursors /*[n_nodes*2], zeroed*/,
                    hipStream_t s);

// ---- RL-sized steps: the whole preparation in ONE launch (small_prep.hip) ----
// Split candidates (quantile: <= 4096 rows, `cum` = the device copy of the target ranks; uniform: <= 8192 rows) and class codes of every
// numeric feature from the row-major matrix, and -- when `want_stats` and the shape qualifies (small_stats' limits) -- the gradient statistics,
// scales and quantised gradients (*stats_done).  false: nothing was launched (the caller runs the separate kernels).
bool small_prep(const float *obs, int N, int F, int B, bool uniform, const int64_t *cum, float *thr, uint32_t *thr_keys, uint16_t *codes,
                uint16_t *codes_fm /*[F][N]: a second, feature-major copy of the numeric codes for kern::small_grow*/, const float *grads, int D, bool centred, int chunk_rows, double *stat, float *meanden, StepScales *sc, int32_t *qg,
                bool want_stats, bool *stats_done, hipStream_t s);

// ---- RL-sized steps: the whole growth of one tree in ONE launch (small_grow.hip) ----
// Per level: LDS histograms of the owned feature slots, scores, arg-max, ONE grid barrier, row routing; then the leaf sums.  The host reads
// `status[0] == seq` (pinned), then MD result blocks of small_grow_res_stride(MD) bytes -- [best_idx i32 x mf][best_score f32 x mf]
// [counts i64 x 4 x mf: total | right | - | -][winner threshold f32 x mf], mf = 2^(MD-1) -- and acc[node id][D+1] of every leaf.
struct SmallGrowIO {
    const uint16_t *codes; const int32_t *qg; const float *grads; const StepScales *scales;
    const uint16_t *codes_fm = nullptr;   // nullable: feature-major copy [n_fm][N] of the codes of the slots < n_fm (written by kern::small_prep)
    int n_fm = 0;
    int n_thr_slots = 0;                  // numeric slots (the rows of thr)
    const FeatureSlot *slots; const float *thr; const float *cand_w; const int32_t *cand_ref;
    int N, D, B, n_slots, NB, MD, min_data;
    bool cosine, oblivious;
    int G;                 // blocks: small_grow_blocks(n_slots)
    void *bests;           // device scratch, small_grow_bests_bytes(MD, G, oblivious)
    unsigned *sync;        // device, 4096 bytes, zero before the first launch (the kernel hands them back zeroed)
    char *res;             // pinned, device-mapped
    char *res_dev;         // device staging of the result blocks, MD * small_grow_res_stride(MD) bytes
    int64_t *acc;          // pinned, device-mapped, [2 << MD][D + 1]
    uint32_t *status;      // pinned, device-mapped, 16 words: sequence word | levels | node count | error (2 = near-tie: level loop) | levels replayed in the kernel
    uint32_t seq;
    uint32_t *prof = nullptr;   // measurement: pinned, 16 words (block 0's time per phase, 10 ns units)
    StepScales *scales_out = nullptr;   // pinned, device-mapped: the step's scales for the host (what publish_pair hands over in the level loop)
    void *near_scratch = nullptr;   // device, small_grow_near_bytes(G, N): the kernel replays a near-tie at one node of a greedy level itself
    const float *meanden = nullptr; // L2: the step's standardisation (mean | std + 1e-8f) for that replay; nullptr: raw gradients (Cosine)
    bool resume = false;        // with replay: continue from the checkpoint the default variant left (status word 5 = 1) instead of growing from the root
    bool replay = false;        // the kernel variant that replays a flagged node itself (launched for a tree the default variant gave up: status word 3 = 2)
    float near_rel = 0.0f;      // > 0: near-tie detection -- the kernel gives the tree up (status word 3 = 2) at the first level whose runner-up
                                // is within near_rel of the winner; the level loop then grows it with the replay (neartie.hip)
};
bool small_grow_supported(int N, int D, int NB, int MD, int n_slots, int n_cand);
int small_grow_blocks(int n_slots);
size_t small_grow_bests_bytes(int MD, int G, bool oblivious);
size_t small_grow_res_stride(int MD);
size_t small_grow_near_bytes(int G, int N, int MD);
bool small_grow(const SmallGrowIO &io, hipStream_t s);   // false: nothing was launched

// ---- leaf values (A11) ----
void leaf_sums(const float *grads, int D, const int32_t *rows, const Chunk *chunks, int n_chunks, const StepScales *sc,
               int64_t *acc /*[n_leaves][D+1], zeroed*/, hipStream_t s);

// ---- prediction (A13) ----
struct PredictModel {
    const int32_t *tree_indices, *depths, *feature_indices, *cat_ids;
    const float *feature_values, *values, *bias;
    const uint8_t *is_numerics, *inequality_directions;
    int n_trees, n_leaves, max_depth, D, oblivious;
    int n_opts;
    const int32_t *opt_start, *opt_stop;
    const float *opt_lr;
    // fast oblivious path (k_predict_obl): per split row 2*max_depth ints (feature index | ~categorical index, threshold bits |
    // category id), and -- when the optimisers' output ranges do not overlap -- one learning rate per output
    const int32_t *cond_pack;
    // fast greedy path (k_predict_grd): every greedy tree rebuilt as a binary tree from its leaves' paths.  Node = int4
    // (feature | ~categorical feature, threshold bits | category id, left child, right child); a child >= 0 is a node of the
    // same tree, < 0 encodes ~(leaf index within the tree).  grd_node_off[t] .. grd_node_off[t+1] are tree t's nodes (root
    // first); grd_ok == 0 when some tree is not a proper binary tree (or has depth 0, Q7): the generic walk is used then.
    const int32_t *grd_nodes;
    const int32_t *grd_node_off;
    int grd_ok, grd_max_nodes, grd_max_leaves;
    int obl_ok, coef_ok;
    uint64_t coef_cover;
    float coef[64];
    float *partial;          // scratch for tree-split prediction of small batches (nullable), partial_floats elements
    size_t partial_floats;
    int tree_chunk;          // set by kern::predict: trees per block column (0 = every block walks the whole range)
    int tree_splits;         // set by kern::predict: block columns (the last one takes the remainder of the range)
    int par_th;              // the model's par_th (the reference's rows / trees per host thread, utils.h:64-80)
    int32_t *slots;          // scratch of the two-launch chain path (predict_chain.hip), slot_ints elements; nullptr: path not offered
    size_t slot_ints;
    // second-generation oblivious path (predict_obl2.hip): leaf values pre-swizzled per tree as [worker 0..3][leaf < 2^max_depth]
    // [DMAX/4] (DMAX = obl2_padded_outputs(D), zero padded), and per tree 2*obl2_maxd condition words RIGHT-aligned (a tree of depth
    // d < obl2_maxd starts with obl2_maxd - d never-true conditions: feature 0, threshold +inf); obl2_maxd = 0: not available
    const float *values_sw;
    const int32_t *cond_ra;
    int obl2_maxd;
    int cat_dict_size;       // dictionary ids of categorical conditions are 1..cat_dict_size (the kernel packs them in 16 bits)
    // packed-code path (predict_reg.hip, k_pack_codes + k_predict_pc): the ensemble's code book, built by the engine when the shape
    // qualifies (nullptr otherwise).  A row becomes pc_row_words dwords: pc_wn numeric words (two 16-bit fields: 0xffff - number of
    // the feature's sorted distinct thresholds below x), then categorical words (one inverted one-hot bit per mentioned category).
    const int32_t *pc_cond;       // per tree obl2_maxd x (word, shift, T), right-aligned like cond_ra (padding: 0, 0, 0 = never true)
    const float *pc_thr;          // sorted distinct thresholds, feature by feature
    const int32_t *pc_thr_off;    // [F + 1]
    const int32_t *pc_cat_slot;   // [cat_dict_size + 1]: dictionary id -> bit slot (-1: none; id 0 = unknown cell)
    const int32_t *pc_word_cols;  // [2 x categorical words]: first and last + 1 column with a slot in the word
    uint32_t *pc_rows;            // scratch [n][pc_row_words]
    int pc_wn, pc_nw, pc_row_words, pc_iters;
};
int obl2_padded_outputs(int D);     // 4, 8, 16, 32, 64 (0: D > 64)
int obl2_levels(int max_depth);     // 4, 6, 8 (0: max_depth > 8)
bool obl2_feasible(int max_depth, int D, bool greedy);   // false: no launch plan can take the shape (the mirror is then not built)
// Small / medium batches against large ensembles: leaf search spread over the chip, then one fused multiply-add chain per (row,
// output) in tree order -- the bits of the one-chain-per-row kernels (predict_chain.hip).  false: not covered, nothing was launched.
size_t predict_chain_slot_ints(int n, int trees);
bool predict_chain(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree, int stop_tree,
                   float *out, hipStream_t s);
bool predict_obl2(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree, int stop_tree,
                  float *out, hipStream_t s);   // false: shape not covered, nothing was launched
// Small greedy ensembles over large batches (predict_grd_stream.hip, round 6): the whole ensemble in LDS, one barrier-free pipeline per
// wave with the next row tile held in registers.  Same mirror and the same bits as predict_obl2's greedy mode; false: not covered.
bool predict_grd_stream(const PredictModel &pm, const float *obs, int F, int Fc, int n, int start_tree, int stop_tree, float *out, hipStream_t s);
// Third-generation oblivious path for large batches (predict_reg.hip): the row tile lives in a bank of VGPRs, every level is a
// VGPR-relative compare, the only LDS traffic is the leaf-value gather.  Same mirrors as predict_obl2; false: not covered.
bool predict_reg(const PredictModel &pm, const float *obs, int F, int Fc, int n, int start_tree, int stop_tree, float *out, hipStream_t s);
// The same walk over PACKED rows (categorical columns, rows wider than the fp32 bank, feature counts that are not multiples of 4).
bool predict_pc(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree, int stop_tree, float *out,
                hipStream_t s);
int predict_pc_bank_words();                      // dwords of a packed row the kernel can hold
bool predict_pc_shape_ok(int obl2_maxd, int D);   // levels / outputs the packed-code kernels are compiled for
// Dictionary encoding of categorical cells on the device (predict): cells [n][Fc][128 B]; the dictionary holds, per
// categorical feature f, its entries sorted by hash: feat_off[f] .. feat_off[f+1] index dict_hash / dict_id / dict_words
// (16 uint64 per entry, normalised).  codes[i*Fc+f] = id of the matching entry, 0 when the cell is not in the dictionary.
// Distinct categorical cells of a batch (A5 on the device).  cat_distinct_insert fills per-feature open-addressing tables
// (keys = raw 128-byte hash, first = smallest row that holds the key) and appends the table slot of every NEW key to list_slot
// (counter = number of distinct (feature, cell) pairs); cat_distinct_verify confirms that every cell equals the cell of its
// key's first row (flags[1] = 1 on a 64-bit hash collision); cat_publish writes header + (feature, first row, hash, cell) of the
// first `cap` list records into mapped pinned host memory.  flags[0] = 1 when a table or the list overflowed.  cat_step_codes
// writes the class of every cell (dictionary sorted by
// hash per feature, id = class, words = raw cell) straight into the grouped u16 code array (slot F+f).
void cat_distinct_insert(const char *cells, int n, int Fc, uint64_t *keys, int32_t *first, int log2_cap, int32_t *flags, int32_t *list_slot,
                         int32_t *counter, int list_cap, hipStream_t s);
void cat_distinct_verify(const char *cells, int n, int Fc, const uint64_t *keys, const int32_t *first, int log2_cap, int32_t *flags,
                         hipStream_t s);
void cat_publish(int32_t *meta, const int32_t *list_slot, const uint64_t *keys, const int32_t *first, int log2_cap, const char *cells,
                 int Fc, int cap, int32_t *h_hdr, int32_t *h_feat, int32_t *h_first, uint64_t *h_hash, char *h_names, int32_t *slot_q,
                 uint32_t seq, hipStream_t s);
// class codes of a step batch from the scan's own tables: cls_of_q[slot_q[slot of the cell]] (see k_cat_step_codes_table)
void cat_step_codes_table(const char *cells, int n, int Fc, int F, const uint64_t *keys, const int32_t *slot_q, const int32_t *cls_of_q,
                          int log2_cap, uint16_t *codes, hipStream_t s);
void cat_step_codes(const char *cells, int n, int Fc, int F, const int32_t *feat_off, const uint64_t *dict_hash, const int32_t *dict_cls,
                    const uint64_t *dict_words, uint16_t *codes, hipStream_t s);
void encode_categories(const char *cells, int n, int Fc, const int32_t *feat_off, const uint64_t *dict_hash, const int32_t *dict_id,
                       const uint64_t *dict_words, int32_t *codes, hipStream_t s);
void predict(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree,
             int stop_tree, float *out, hipStream_t s);

// ---- Linear TreeSHAP (shap.hip): a uniform program over explicit trees, one thread per (sample, output) ----
enum { SHAP_ENTER = 0, SHAP_AFTER_LEFT = 1, SHAP_AFTER_RIGHT = 2, SHAP_EXIT = 3 };
struct ShapOp { int32_t kind, node, level, via; };   // via: feature tested by the parent (-1 at a root)
struct ShapNodeRec {                                  // explain.h ShapNode, flattened; node indices are global over the program
    int32_t feature, flags /*1 numeric, 2 tied to parent, 4 leaf, 8 right child*/, cat_id, pred;
    int32_t n_unique, n_unique_parent, deg_left, deg_right;
    float threshold, weight, weight_parent, pad;
};
int shap_block_threads(int max_depth, int D);   // 0: this (max_depth, D) does not fit the kernel (LDS) -- evaluate on the host
// out [n_samples][n_num + n_cat][D] is accumulated into (zero it first)
void shap_values(const ShapOp *ops, int n_ops, const ShapNodeRec *nodes, const float *leaf_value, const float *obs, int n_num,
                 const int32_t *cat_ids, int n_cat, int n_samples, int D, int max_depth, const float *norm, const float *base,
                 const float *offset, float *out, hipStream_t s);

}  // namespace kern
}  // namespace gbrl
