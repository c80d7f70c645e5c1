// Host-only check of gbrl_amd/csrc/hash_order_replay.h against the real libstdc++ container (compiled and run by tests/test_host.py).
#include "hash_order_replay.h"
#include <cstdio>
#include <random>
#include <string>
struct K { int id; std::size_t h; };
struct KH { std::size_t operator()(const K &k) const noexcept { return k.h; } };
struct KE { bool operator()(const K &a, const K &b) const noexcept { return a.id == b.id; } };
int main() {
    std::mt19937_64 rng(7);
    int bad = 0, cases = 0;
    for (int rep = 0; rep < 400; ++rep) {
        const int n = rep < 50 ? rep : static_cast<int>(rng() % 6000);
        const int mode = rep % 4;
        std::vector<std::size_t> h(n);
        for (auto &x : h) x = mode == 0 ? rng() : mode == 1 ? rng() % 97 : mode == 2 ? (rng() % 5) * 1109 : rng() % (n + 1);
        std::unordered_map<K, int, KH, KE> m;
        for (int i = 0; i < n; ++i) m.emplace(K{i, h[i]}, i);
        std::vector<int> want;
        for (auto &kv : m) want.push_back(kv.second);
        bad += want != gbrl::libstdcxx_unique_insert_order(h);
        ++cases;
    }
    // and against string keys with their real std::hash (what the reference's container holds)
    for (int rep = 0; rep < 100; ++rep) {
        const int n = static_cast<int>(rng() % 3000);
        std::unordered_map<std::string, int> m;
        std::vector<std::size_t> h;
        for (int i = 0; i < n; ++i) {
            std::string key(128, '\0');
            const std::string t = "tok" + std::to_string(rng() % 100000) + "x" + std::to_string(i);
            key.replace(0, t.size(), t);
            key += "_" + std::to_string(rng() % 64);
            h.push_back(std::hash<std::string>{}(key));
            m.emplace(key, i);
        }
        std::vector<int> want;
        for (auto &kv : m) want.push_back(kv.second);
        bad += want != gbrl::libstdcxx_unique_insert_order(h);
        ++cases;
    }
    std::printf("%d cases, %d mismatches\n", cases, bad);
    return bad ? 1 : 0;
}
