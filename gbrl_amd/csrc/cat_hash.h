// cat_hash.h -- one 64-bit hash of a 128-byte categorical cell, identical on host and device.
// strcmp semantics (node.cpp:75, predictor.cpp:215): only the bytes before the first NUL count, so the cell is normalised
// (everything from the first NUL on is treated as zero) while it is hashed.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define GBRL_HD __host__ __device__ __forceinline__
#else
#define GBRL_HD inline
#endif

namespace gbrl {

// words: the cell as 16 little-endian uint64.  Returns the hash; norm (optional) receives the normalised words.
GBRL_HD uint64_t cat_cell_hash(const uint64_t *words, uint64_t *norm /*nullable [16]*/) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    bool ended = false;
    for (int k = 0; k < 16; ++k) {
        uint64_t w = ended ? 0ull : words[k];
        if (!ended) {
            // first zero byte of w (if any): classic has-zero-byte test, then keep only the bytes below it
            const uint64_t z = (w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull;
            if (z) {
                int byte = 0;
                while (((w >> (8 * byte)) & 0xffull) != 0) ++byte;
                w = byte ? (w & ((1ull << (8 * byte)) - 1ull)) : 0ull;
                ended = true;
            }
        }
        if (norm) norm[k] = w;
        h ^= w + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xFF51AFD7ED558CCDull;
        h ^= h >> 33;
    }
    return h;
}

// Hash of the RAW 128 bytes (no normalisation): the reference keys its candidate map by std::string(cell, 128)
// (split_candidate_generator.cpp:121), so cells that differ only behind the first NUL are different candidates there.
GBRL_HD uint64_t cat_cell_hash_raw(const uint64_t *words) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < 16; ++k) {
        h ^= words[k] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xFF51AFD7ED558CCDull;
        h ^= h >> 33;
    }
    return h ? h : 1ull;   // 0 marks an empty hash-table slot
}

}  // namespace gbrl
