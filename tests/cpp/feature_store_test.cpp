// Randomised check of cg::MapServer (csrc/host/feature_store.h) against the reference's containers:
// std::map<FeatureIDType, std::map<StateIDType, Vector4>> driven through the same operations the filter performs.
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include "../../msckf_stereo_c_amd/csrc/host/feature_store.h"

using namespace cg;
typedef std::map<StateIDType, std::array<double, 4>> Obs;

int main() {
    std::mt19937_64 rng(12345);
    MapServer ms;
    ms.set_clone_rows(13);
    std::map<FeatureIDType, Obs> ref;
    std::vector<StateIDType> window;          // live clone ids, ascending
    std::vector<int> window_row;              // table row of each
    std::vector<int> free_rows;
    for (int r = 12; r >= 0; --r) free_rows.push_back(r);
    StateIDType next_state = 0;
    FeatureIDType next_feat = 0;
    std::vector<FeatureIDType> live;
    long long checks = 0;
    for (int frame = 0; frame < 3000; ++frame) {
        // new clone
        window.push_back(next_state++); window_row.push_back(free_rows.back()); free_rows.pop_back();
        const int kn = (int)window.size() - 1;
        // observations: most live features again, some new ones, sometimes a stale id that was erased long ago
        std::vector<FeatureIDType> seen;
        for (FeatureIDType f : live) if (rng() % 10 != 0) seen.push_back(f);
        const int n_new = (int)(rng() % 12);
        for (int i = 0; i < n_new; ++i) seen.push_back(next_feat++);
        if (rng() % 7 == 0 && next_feat > 50) seen.push_back((FeatureIDType)(rng() % (uint64_t)(next_feat - 40)));
        if (rng() % 5 == 0 && !seen.empty()) seen.push_back(seen[rng() % seen.size()]);      // a duplicate record (Q1)
        for (size_t i = seen.size(); i > 1; --i) std::swap(seen[i - 1], seen[rng() % i]);       // message order is not id order
        for (FeatureIDType f : seen) {
            std::array<double, 4> z = {(double)(rng() % 1000), (double)frame, (double)f, 1.0};
            ref[f][window.back()] = z;
            bool created;
            const int s = ms.find_or_add(f, created);
            double *zz = ms.z(window_row[kn], s);
            for (int k = 0; k < 4; ++k) zz[k] = z[k];
            ms.mask(s) |= 1ULL << kn;
        }
        live = seen;
        // lost features: not seen in the newest clone -> erased
        std::vector<size_t> ranks;
        for (size_t r = 0; r < ms.size(); ++r) if (!(ms.mask(ms.slot_at(r)) & (1ULL << kn))) ranks.push_back(r);
        ms.erase_ranks(ranks);
        for (auto it = ref.begin(); it != ref.end();) { if (it->second.find(window.back()) == it->second.end()) it = ref.erase(it); else ++it; }
        // prune two clones when the window is full
        if (window.size() >= 12) {
            int a = (int)(rng() % (window.size() - 1)), b = a + 1 + (int)(rng() % (window.size() - 1 - a));
            for (auto &kv : ref) { kv.second.erase(window[a]); kv.second.erase(window[b]); }
            ms.remove_clone_bit(b); ms.remove_clone_bit(a);
            free_rows.push_back(window_row[b]); free_rows.push_back(window_row[a]);
            window.erase(window.begin() + b); window_row.erase(window_row.begin() + b);
            window.erase(window.begin() + a); window_row.erase(window_row.begin() + a);
        }
        // compare everything
        if (ms.size() != ref.size()) { std::printf("frame %d: size %zu vs %zu\n", frame, ms.size(), ref.size()); return 1; }
        size_t r = 0;
        for (const auto &kv : ref) {
            if (ms.id_at(r) != kv.first) { std::printf("frame %d: rank %zu id %lld vs %lld\n", frame, r, ms.id_at(r), kv.first); return 1; }
            const int s = ms.slot_at(r);
            if (ms.find(kv.first) != s) { std::printf("frame %d: hash lookup of %lld\n", frame, kv.first); return 1; }
            uint64_t m = ms.mask(s);
            if ((size_t)__builtin_popcountll(m) != kv.second.size()) { std::printf("frame %d: obs count of %lld\n", frame, kv.first); return 1; }
            auto it = kv.second.begin();
            for (uint64_t bb = m; bb; bb &= bb - 1, ++it) {
                const int k = __builtin_ctzll(bb);
                if (window[k] != it->first) { std::printf("frame %d: obs state id\n", frame); return 1; }
                const double *z = ms.z(window_row[k], s);
                for (int q = 0; q < 4; ++q) if (z[q] != it->second[q]) { std::printf("frame %d: obs value\n", frame); return 1; }
                ++checks;
            }
            ++r;
        }
        if (ms.find(next_feat + 5) != -1) { std::printf("phantom id\n"); return 1; }
    }
    std::printf("ok %lld observation checks\n", checks);
    return 0;
}
