// categorical.hip -- categorical cells on the device: distinct (feature, cell) pairs of a step batch (A5), class codes of the
// step's candidates, dictionary ids of predict batches.  Cells are 128-byte strings (MAX_CHAR_SIZE, types.h:55-58).
#include "kernels.h"
#include "kernels_common.h"
#include "cat_hash.h"

#include <algorithm>

namespace gbrl {
namespace kern {

// ------------------------------------------------------------------------------------------------------------
// A5 on the device: the distinct (feature, cell) pairs of a batch and their first rows, through per-feature open-addressing
// hash tables keyed by the hash of the RAW 128 bytes.  The host only sees the distinct cells (a few hundred), inserts them
// into the reference's container in the reference's insertion order (feature-major, first occurrence) and hands the
// candidate dictionary back for k_cat_step_codes.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t load_cell_raw_hash(const char *cell, uint64_t (&w)[16]) {
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(cell);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const ulonglong2 v = src[k]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
    return cat_cell_hash_raw(w);
}
constexpr int kCatMaxProbes = 512;
__global__ __launch_bounds__(256) void k_cat_distinct_insert(const char *__restrict__ cells, size_t n_cells, int Fc,
                                                             unsigned long long *__restrict__ keys, int32_t *__restrict__ first,
                                                             int log2_cap, int32_t *__restrict__ flags, int32_t *__restrict__ list_slot,
                                                             int32_t *__restrict__ counter, int list_cap) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const int f = static_cast<int>(i % Fc), row = static_cast<int>(i / Fc);
    uint64_t w[16];
    const uint64_t h = load_cell_raw_hash(cells + i * 128, w);
    const uint32_t mask = (1u << log2_cap) - 1u;
    uint32_t slot = static_cast<uint32_t>(h >> 20) & mask;
    const size_t base = static_cast<size_t>(f) << log2_cap;
    for (int p = 0; p < kCatMaxProbes; ++p) {
        const unsigned long long old = atomicCAS(&keys[base + slot], 0ull, static_cast<unsigned long long>(h));
        if (old == 0ull || old == h) {
            atomicMin(&first[base + slot], row);
            if (old == 0ull && list_slot) {   // this thread created the entry: one list record per distinct (feature, cell)
                const int idx = atomicAdd(counter, 1);
                if (idx < list_cap) list_slot[idx] = static_cast<int32_t>(base + slot);
                else flags[0] = 1;
            }
            return;
        }
        slot = (slot + 1) & mask;
    }
    flags[0] = 1;   // table too full
}
// Everything the host needs about the distinct cells, written into MAPPED PINNED host memory by one launch (no copy-engine
// transfers, one synchronisation): header {table overflow, hash collision, distinct count, records published}, then per record
// the feature, the first row, the raw hash and the 128 bytes of the cell itself (gathered from its first row).
// slot_q (nullable): list index of every live table slot, for k_cat_step_codes_table.  The LAST block to finish stores `seq` to
// h_hdr[4] (system scope): the host polls that word instead of waiting for an event or the stream (meta[3] counts the blocks).
__global__ __launch_bounds__(256) void k_cat_publish(int32_t *__restrict__ meta, const int32_t *__restrict__ list_slot,
                                                     const unsigned long long *__restrict__ keys, const int32_t *__restrict__ first,
                                                     int log2_cap, const char *__restrict__ cells, int Fc, int cap,
                                                     int32_t *__restrict__ h_hdr, int32_t *__restrict__ h_feat, int32_t *__restrict__ h_first,
                                                     unsigned long long *__restrict__ h_hash, char *__restrict__ h_names,
                                                     int32_t *__restrict__ slot_q, uint32_t seq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte piece per thread
    const int n = min(meta[2], cap);
    if (i == 0) { h_hdr[0] = meta[0]; h_hdr[1] = meta[1]; h_hdr[2] = meta[2]; h_hdr[3] = n; }
    if (i < n * 8) {
        const int item = i >> 3, piece = i & 7;
        const int32_t slot = list_slot[item];
        const int feat = slot >> log2_cap, row = first[slot];
        if (piece == 0) {
            h_feat[item] = feat; h_first[item] = row; h_hash[item] = keys[slot];
            if (slot_q) slot_q[slot] = item;
        }
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(cells + (static_cast<size_t>(row) * Fc + feat) * 128);
        reinterpret_cast<ulonglong2 *>(h_names + static_cast<size_t>(item) * 128)[piece] = src[piece];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&meta[3], 1) == static_cast<int>(gridDim.x) - 1) {
        meta[3] = 0;     // ready for a second publish of the same scan (launches on one stream do not overlap)
        __hip_atomic_store(reinterpret_cast<uint32_t *>(h_hdr) + 4, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(256) void k_cat_distinct_verify(const char *__restrict__ cells, size_t n_cells, int Fc,
                                                             const unsigned long long *__restrict__ keys, const int32_t *__restrict__ first,
                                                             int log2_cap, int32_t *__restrict__ flags) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const int f = static_cast<int>(i % Fc);
    uint64_t w[16];
    const uint64_t h = load_cell_raw_hash(cells + i * 128, w);
    const uint32_t mask = (1u << log2_cap) - 1u;
    uint32_t slot = static_cast<uint32_t>(h >> 20) & mask;
    const size_t base = static_cast<size_t>(f) << log2_cap;
    for (int p = 0; p < kCatMaxProbes; ++p) {
        const unsigned long long k = keys[base + slot];
        if (k == h) {
            const uint64_t *rep = reinterpret_cast<const uint64_t *>(cells + (static_cast<size_t>(first[base + slot]) * Fc + f) * 128);
            bool same = true;
#pragma unroll
            for (int q = 0; q < 16; ++q) same &= rep[q] == w[q];
            if (!same) flags[1] = 1;   // two different cells share a 64-bit hash
            return;
        }
        if (k == 0ull) break;
        slot = (slot + 1) & mask;
    }
    flags[0] = 1;
}
__global__ __launch_bounds__(256) void k_cat_step_codes(const char *__restrict__ cells, size_t n_cells, int n, int Fc, int F,
                                                        const int32_t *__restrict__ feat_off, const uint64_t *__restrict__ dict_hash,
                                                        const int32_t *__restrict__ dict_cls, const uint64_t *__restrict__ dict_words,
                                                        uint16_t *__restrict__ codes) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const int f = static_cast<int>(i % Fc);
    const size_t r = i / Fc;
    uint64_t w[16];
    const uint64_t h = load_cell_raw_hash(cells + i * 128, w);
    int lo = feat_off[f], hi = feat_off[f + 1];
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (dict_hash[mid] < h) lo = mid + 1; else hi = mid; }
    int cls = 0;
    for (int e = lo; e < feat_off[f + 1] && dict_hash[e] == h; ++e) {
        bool same = true;
        for (int k = 0; k < 16; ++k) same &= dict_words[static_cast<size_t>(e) * 16 + k] == w[k];
        if (same) { cls = dict_cls[e]; break; }
    }
    const int slot = F + f;
    codes[(static_cast<size_t>(slot / kCodeGroup) * n + r) * kCodeGroup + (slot % kCodeGroup)] = static_cast<uint16_t>(cls);
}
void cat_distinct_insert(const char *cells, int n, int Fc, uint64_t *keys, int32_t *first, int log2_cap, int32_t *flags, int32_t *list_slot,
                         int32_t *counter, int list_cap, hipStream_t s) {
    const size_t n_cells = static_cast<size_t>(n) * Fc;
    if (!n_cells) return;
    hipLaunchKernelGGL(k_cat_distinct_insert, dim3(static_cast<unsigned>((n_cells + 255) / 256)), dim3(256), 0, s, cells, n_cells, Fc,
                       reinterpret_cast<unsigned long long *>(keys), first, log2_cap, flags, list_slot, counter, list_cap);
}
void cat_publish(int32_t *meta, const int32_t *list_slot, const uint64_t *keys, const int32_t *first, int log2_cap, const char *cells,
                 int Fc, int cap, int32_t *h_hdr, int32_t *h_feat, int32_t *h_first, uint64_t *h_hash, char *h_names, int32_t *slot_q,
                 uint32_t seq, hipStream_t s) {
    hipLaunchKernelGGL(k_cat_publish, dim3((std::max(1, cap) * 8 + 255) / 256), dim3(256), 0, s, meta, list_slot,
                       reinterpret_cast<const unsigned long long *>(keys), first, log2_cap, cells, Fc, cap, h_hdr, h_feat, h_first,
                       reinterpret_cast<unsigned long long *>(h_hash), h_names, slot_q, seq);
}
// Class codes of a step batch straight from the scan's own hash tables (one GPU, ordinary steps): the batch's cells are all in
// the table and k_cat_distinct_verify has shown that equal hashes mean equal cells IN THIS BATCH, so the probe that inserted a cell finds
// it again; cls_of_q[list index] is the only thing the host hands back (8 KiB at configs[4] instead of a 270 KiB dictionary).
__global__ __launch_bounds__(256) void k_cat_step_codes_table(const char *__restrict__ cells, size_t n_cells, int n, int Fc, int F,
                                                              const unsigned long long *__restrict__ keys, const int32_t *__restrict__ slot_q,
                                                              const int32_t *__restrict__ cls_of_q, int log2_cap, uint16_t *__restrict__ codes) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const int f = static_cast<int>(i % Fc);
    const size_t r = i / Fc;
    uint64_t w[16];
    const uint64_t h = load_cell_raw_hash(cells + i * 128, w);
    const uint32_t mask = (1u << log2_cap) - 1u;
    uint32_t slot = static_cast<uint32_t>(h >> 20) & mask;
    const size_t base = static_cast<size_t>(f) << log2_cap;
    int cls = 0;
    for (int p = 0; p < kCatMaxProbes; ++p) {
        const unsigned long long k = keys[base + slot];
        if (k == h) { cls = cls_of_q[slot_q[base + slot]]; break; }
        if (k == 0ull) break;
        slot = (slot + 1) & mask;
    }
    const int cs = F + f;
    codes[(static_cast<size_t>(cs / kCodeGroup) * n + r) * kCodeGroup + (cs % kCodeGroup)] = static_cast<uint16_t>(cls);
}
void cat_step_codes_table(const char *cells, int n, int Fc, int F, const uint64_t *keys, const int32_t *slot_q, const int32_t *cls_of_q,
                          int log2_cap, uint16_t *codes, hipStream_t s) {
    const size_t n_cells = static_cast<size_t>(n) * Fc;
    if (!n_cells) return;
    hipLaunchKernelGGL(k_cat_step_codes_table, dim3(static_cast<unsigned>((n_cells + 255) / 256)), dim3(256), 0, s, cells, n_cells, n, Fc, F,
                       reinterpret_cast<const unsigned long long *>(keys), slot_q, cls_of_q, log2_cap, codes);
}
void cat_distinct_verify(const char *cells, int n, int Fc, const uint64_t *keys, const int32_t *first, int log2_cap, int32_t *flags,
                         hipStream_t s) {
    const size_t n_cells = static_cast<size_t>(n) * Fc;
    if (!n_cells) return;
    hipLaunchKernelGGL(k_cat_distinct_verify, dim3(static_cast<unsigned>((n_cells + 255) / 256)), dim3(256), 0, s, cells, n_cells, Fc,
                       reinterpret_cast<const unsigned long long *>(keys), first, log2_cap, flags);
}
void cat_step_codes(const char *cells, int n, int Fc, int F, const int32_t *feat_off, const uint64_t *dict_hash, const int32_t *dict_cls,
                    const uint64_t *dict_words, uint16_t *codes, hipStream_t s) {
    const size_t n_cells = static_cast<size_t>(n) * Fc;
    if (!n_cells) return;
    hipLaunchKernelGGL(k_cat_step_codes, dim3(static_cast<unsigned>((n_cells + 255) / 256)), dim3(256), 0, s, cells, n_cells, n, Fc, F,
                       feat_off, dict_hash, dict_cls, dict_words, codes);
}

// One thread per (row, categorical feature): normalise + hash the 128-byte cell, binary-search the feature's hash-sorted
// dictionary, confirm with a full comparison of the normalised words.
__global__ __launch_bounds__(256) void k_encode_categories(const char *__restrict__ cells, size_t n_cells, int Fc,
                                                           const int32_t *__restrict__ feat_off, const uint64_t *__restrict__ dict_hash,
                                                           const int32_t *__restrict__ dict_id, const uint64_t *__restrict__ dict_words,
                                                           int32_t *__restrict__ codes) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const int f = static_cast<int>(i % Fc);
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(cells + i * 128);
    uint64_t w[16], nw[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const ulonglong2 v = src[k]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
    const uint64_t h = cat_cell_hash(w, nw);
    int lo = feat_off[f], hi = feat_off[f + 1];
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (dict_hash[mid] < h) lo = mid + 1; else hi = mid; }
    int code = 0;
    for (int e = lo; e < feat_off[f + 1] && dict_hash[e] == h; ++e) {   // equal hashes (practically one entry): compare the words
        bool same = true;
        for (int k = 0; k < 16; ++k) same &= dict_words[static_cast<size_t>(e) * 16 + k] == nw[k];
        if (same) { code = dict_id[e]; break; }
    }
    codes[i] = code;
}
void encode_categories(const char *cells, int n, int Fc, const int32_t *feat_off, const uint64_t *dict_hash, const int32_t *dict_id,
                       const uint64_t *dict_words, int32_t *codes, hipStream_t s) {
    const size_t n_cells = static_cast<size_t>(n) * Fc;
    if (n_cells == 0) return;
    hipLaunchKernelGGL(k_encode_categories, dim3(static_cast<unsigned>((n_cells + 255) / 256)), dim3(256), 0, s, cells, n_cells, Fc, feat_off,
                       dict_hash, dict_id, dict_words, codes);
}

}  // namespace kern
}  // namespace gbrl
