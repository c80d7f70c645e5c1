// hash_order_replay.h -- the iteration order of a libstdc++ std::unordered_map after a sequence of insertions of DISTINCT keys,
// computed from the keys' hash values alone.
//
// The reference reads its categorical split candidates off the iteration order of a std::unordered_map<std::string, ...>
// (split_candidate_generator.cpp:117-163, Q8).  That order is an implementation detail of the container -- a function of the keys'
// std::hash values and of the insertion sequence -- so the engine replays the container's algorithm instead of building the container:
// bits/hashtable.h _M_insert_unique_node -> (_M_rehash -> _M_rehash_aux(unique keys)) -> _M_insert_bucket_begin, restated on index
// arrays and driven by the library's own std::__detail::_Prime_rehash_policy object, so the bucket-count sequence is the container's
// by construction.  tests/test_host.py compiles this header against the real container on random hash sequences; the GPU test suite
// runs every categorical step with GBRL_HIP_CAT_CHECK=1, which replays the string-keyed container beside it.
#pragma once

#include <cstddef>
#include <unordered_map>
#if !defined(__GLIBCXX__)
#error "hash_order_replay.h restates libstdc++'s unordered_map insertion (bits/hashtable.h); build the product with libstdc++, the reference's standard library"
#endif
#include <utility>
#include <vector>

namespace gbrl {

// hashes[k] = std::hash of the k-th key inserted (all keys distinct).  Returns the insertion indices in iteration order.
inline std::vector<int> libstdcxx_unique_insert_order(const std::vector<std::size_t> &hashes) {
    std::__detail::_Prime_rehash_policy pol;
    // "before" pointers: -1 none, -2 the before-begin sentinel, k >= 0 node k
    std::vector<int> bucket(1, -1), nb, next;
    next.reserve(hashes.size());
    int head = -1;
    std::size_t n_bkt = 1;
    for (std::size_t node = 0; node < hashes.size(); ++node) {
        const std::pair<bool, std::size_t> grow = pol._M_need_rehash(n_bkt, node, 1);
        if (grow.first) {   // _M_rehash_aux(unique keys): every node is relinked in iteration order
            nb.assign(grow.second, -1);
            int p = head;
            head = -1;
            std::size_t bbegin = 0;
            while (p >= 0) {
                const int nx = next[p];
                const std::size_t bk = hashes[p] % grow.second;
                if (nb[bk] == -1) {
                    next[p] = head; head = p; nb[bk] = -2;
                    if (next[p] >= 0) nb[bbegin] = p;
                    bbegin = bk;
                } else {
                    int &after = nb[bk] == -2 ? head : next[nb[bk]];
                    next[p] = after; after = p;
                }
                p = nx;
            }
            bucket.swap(nb);
            n_bkt = grow.second;
        }
        next.push_back(-1);
        const int me = static_cast<int>(node);
        const std::size_t bk = hashes[node] % n_bkt;
        if (bucket[bk] != -1) {   // _M_insert_bucket_begin: first node of a non-empty bucket ...
            int &after = bucket[bk] == -2 ? head : next[bucket[bk]];
            next[me] = after; after = me;
        } else {                  // ... or new head of the whole list
            next[me] = head; head = me;
            if (next[me] >= 0) bucket[hashes[next[me]] % n_bkt] = me;
            bucket[bk] = -2;
        }
    }
    std::vector<int> order;
    order.reserve(hashes.size());
    for (int p = head; p >= 0; p = next[p]) order.push_back(p);
    return order;
}

}  // namespace gbrl
