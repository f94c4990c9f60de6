// feature_store.h — the filter's feature / observation bookkeeping (reference: MapServer = std::map<FeatureIDType,
// Feature>, Feature::observations = std::map<StateIDType, Vector4>, feature.hpp:139,166-168) as flat tables.
//
// What the reference's maps are used for is narrow: append one observation per tracked feature and frame, find the
// features that were not observed in the newest clone, walk a feature's observations in ascending state id, erase the
// observations of the two clones a pruning step removes, erase features.  With a few hundred live features of up to
// max_cam_state_size observations each, per-feature containers put every one of those passes at a cache miss per
// feature (the host bookkeeping was ~110 us per stream and frame, half of a frame's critical path).  Here:
//   * a feature is a row slot; ids are kept sorted (iteration order of std::map) next to their slots;
//   * which clones observed a feature is a 64-bit mask in CLONE ORDER space: bit k = the k-th oldest clone of the
//     window (max_cam_state_size <= 64).  "Observed in the newest clone", "number of observations", "first / last
//     observation" are single bit operations; removing a clone removes one bit position from every mask;
//   * the observations themselves live in a clone-major table z[clone slot][feature slot][4]: a frame's new
//     observations are one sequential column write, the two clones of a pruning update two sequential column reads.
// Semantics (iteration in ascending id, observations in ascending state id, last write wins, erase) are the maps'.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include "cg_types.h"

namespace cg {

typedef long long int StateIDType;
typedef long long int FeatureIDType;

class MapServer {
  public:
    static constexpr int kMaxClones = 64;

    // rows of the observation table = clone slots in use at once (window size incl. the newest clone); before first use
    void set_clone_rows(int n) { if (n != clone_rows_ && n >= 1 && n <= kMaxClones) { clone_rows_ = n; z_.clear(); row_cap_ = 0; ensure_rows(used_slots_); } }
    int clone_rows() const { return clone_rows_; }
    size_t size() const { return ids_.size(); }
    bool empty() const { return ids_.empty(); }
    void clear() {
        ids_.clear(); slots_.clear(); free_.clear();
        mask_.clear(); pos_.clear(); init_.clear();
        std::fill(hkey_.begin(), hkey_.end(), kEmpty);
        used_slots_ = 0;
    }

    // ---- rank access (rank = position in ascending id order)
    FeatureIDType id_at(size_t rank) const { return ids_[rank]; }
    int slot_at(size_t rank) const { return slots_[rank]; }
    uint64_t mask(int slot) const { return mask_[slot]; }
    uint64_t &mask(int slot) { return mask_[slot]; }
    bool is_initialized(int slot) const { return init_[slot] != 0; }
    const Vector3 &position(int slot) const { return pos_[slot]; }
    void set_position(int slot, const Vector3 &p) { pos_[slot] = p; init_[slot] = 1; }

    // slot of `id`, or -1
    int find(FeatureIDType id) const {
        if (hkey_.empty()) return -1;
        const size_t m = hkey_.size() - 1;
        for (size_t h = hash(id) & m;; h = (h + 1) & m) {
            if (hkey_[h] == id) return hval_[h];
            if (hkey_[h] == kEmpty) return -1;
        }
    }
    // slot of `id`, created (no observations, not initialised) if absent
    int find_or_add(FeatureIDType id, bool &created) {
        int s = find(id);
        created = s < 0;
        if (!created) return s;
        if (!free_.empty()) { s = free_.back(); free_.pop_back(); }
        else { s = used_slots_++; ensure_rows(used_slots_); }
        mask_[s] = 0; init_[s] = 0; pos_[s] = Vector3();
        hash_insert(id, s);                                                                   // (may rehash from ids_: before the id joins them)
        if (ids_.empty() || ids_.back() < id) { ids_.push_back(id); slots_.push_back(s); }    // ids are counters: an append
        else {                                                                                // (a stale id coming back, Q1)
            const size_t at = (size_t)(std::lower_bound(ids_.begin(), ids_.end(), id) - ids_.begin());
            ids_.insert(ids_.begin() + at, id); slots_.insert(slots_.begin() + at, s);
        }
        return s;
    }
    // erase the features at the given ranks (ascending, unique) in one compaction pass
    void erase_ranks(const std::vector<size_t> &ranks) {
        if (ranks.empty()) return;
        size_t w = ranks[0], q = 0;
        for (size_t r = ranks[0]; r < ids_.size(); ++r) {
            if (q < ranks.size() && ranks[q] == r) { hash_erase(ids_[r]); free_.push_back(slots_[r]); ++q; continue; }
            ids_[w] = ids_[r]; slots_[w] = slots_[r]; ++w;
        }
        ids_.resize(w); slots_.resize(w);
    }

    // ---- observations.  Clone slots are table rows handed out by the caller (0 .. kMaxClones-1).
    double *z(int clone_slot, int slot) { return &z_[((size_t)clone_slot * row_cap_ + (size_t)slot) * 4]; }
    const double *z(int clone_slot, int slot) const { return &z_[((size_t)clone_slot * row_cap_ + (size_t)slot) * 4]; }

    // ---- software prefetch.  A stream's tables are ~1 MB and a host thread walks the tables of hundreds of streams per frame:
    // every per-feature access is a cache miss, and the loops over ranks know their slots ahead of time (slots_ is sequential).
    void prefetch_rank(size_t rank) const {                     // mask / init flag of the feature at `rank`
        if (rank < slots_.size()) { const int s = slots_[rank]; __builtin_prefetch(&mask_[s]); __builtin_prefetch(&init_[s]); }
    }
    void prefetch_obs(size_t rank, int row_a, int row_b) const {    // its observations in two clone rows and its position
        if (rank < slots_.size()) { const int s = slots_[rank]; __builtin_prefetch(z(row_a, s)); __builtin_prefetch(z(row_b, s)); __builtin_prefetch(&pos_[s]); }
    }
    void prefetch_all_obs(size_t rank, uint64_t skip_bit, const std::vector<int> &row_of_order) const {   // every observation of a feature that lost track
        if (rank >= slots_.size()) return;
        const int s = slots_[rank];
        const uint64_t m = mask_[s];
        if ((m & skip_bit) || __builtin_popcountll(m) < 3) return;
        __builtin_prefetch(&pos_[s]);
        for (uint64_t b = m; b; b &= b - 1) __builtin_prefetch(z(row_of_order[__builtin_ctzll(b)], s));
    }
    void prefetch_id(FeatureIDType id) const {                  // the hash bucket of `id`
        if (!hkey_.empty()) { const size_t h = hash(id) & (hkey_.size() - 1); __builtin_prefetch(&hkey_[h]); __builtin_prefetch(&hval_[h]); }
    }
    void prefetch_slot(int clone_slot, int slot) const { __builtin_prefetch(&mask_[slot], 1); __builtin_prefetch(z(clone_slot, slot), 1); }

    // remove bit position k (a clone leaving the window) from every mask: bits above k move down by one
    void remove_clone_bit(int k) {
        const uint64_t low = (k == 0) ? 0ULL : (~0ULL >> (64 - k));
        for (size_t r = 0; r < slots_.size(); ++r) {
            if (r + 16 < slots_.size()) __builtin_prefetch(&mask_[slots_[r + 16]], 1);
            uint64_t &m = mask_[slots_[r]];
            m = (m & low) | ((m >> 1) & ~low);
        }
    }

  private:
    static constexpr FeatureIDType kEmpty = (FeatureIDType)0x8000000000000000ULL;
    static size_t hash(FeatureIDType id) { uint64_t x = (uint64_t)id * 0x9E3779B97F4A7C15ULL; return (size_t)(x >> 20); }
    void ensure_rows(int n) {
        if ((int)mask_.size() < n) { mask_.resize(n, 0); pos_.resize(n); init_.resize(n, 0); }
        if (n > row_cap_) {
            // grow the clone-major table: re-lay every clone column with the new stride
            const int cap = std::max(256, std::max(n, row_cap_ * 2));
            std::vector<double> nz((size_t)clone_rows_ * cap * 4, 0.0);
            if (row_cap_ > 0)
                for (int c = 0; c < clone_rows_; ++c)
                    std::memcpy(&nz[(size_t)c * cap * 4], &z_[(size_t)c * row_cap_ * 4], sizeof(double) * 4 * (size_t)row_cap_);
            z_.swap(nz);
            row_cap_ = cap;
        }
    }
    void hash_insert(FeatureIDType id, int slot) {
        if ((ids_.size() + 1) * 2 > hkey_.size()) rehash(hkey_.empty() ? 1024 : hkey_.size() * 2);
        const size_t m = hkey_.size() - 1;
        size_t h = hash(id) & m;
        while (hkey_[h] != kEmpty) h = (h + 1) & m;
        hkey_[h] = id; hval_[h] = slot;
    }
    void hash_erase(FeatureIDType id) {     // linear probing with backward shift: no tombstones
        const size_t m = hkey_.size() - 1;
        size_t h = hash(id) & m;
        while (hkey_[h] != id) { if (hkey_[h] == kEmpty) return; h = (h + 1) & m; }
        for (size_t j = (h + 1) & m;; j = (j + 1) & m) {
            if (hkey_[j] == kEmpty) break;
            const size_t home = hash(hkey_[j]) & m;
            // the entry at j may move to the hole at h if its home is cyclically outside (h, j]
            const bool outside = (h <= j) ? (home <= h || home > j) : (home <= h && home > j);
            if (outside) { hkey_[h] = hkey_[j]; hval_[h] = hval_[j]; h = j; }
        }
        hkey_[h] = kEmpty;
    }
    void rehash(size_t n) {
        hkey_.assign(n, kEmpty); hval_.assign(n, 0);
        const size_t m = n - 1;
        for (size_t r = 0; r < ids_.size(); ++r) {
            size_t h = hash(ids_[r]) & m;
            while (hkey_[h] != kEmpty) h = (h + 1) & m;
            hkey_[h] = ids_[r]; hval_[h] = slots_[r];
        }
    }

    std::vector<FeatureIDType> ids_;      // ascending
    std::vector<int32_t> slots_;          // slot of ids_[rank]
    std::vector<int32_t> free_;
    int used_slots_ = 0, row_cap_ = 0, clone_rows_ = kMaxClones;
    std::vector<uint64_t> mask_;          // per slot
    std::vector<Vector3> pos_;
    std::vector<uint8_t> init_;
    std::vector<double> z_;               // [clone_rows_][row_cap_][4]
    std::vector<FeatureIDType> hkey_;
    std::vector<int32_t> hval_;
};

}  // namespace cg
