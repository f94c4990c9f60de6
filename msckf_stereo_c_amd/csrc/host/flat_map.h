// flat_map.h — the subset of std::map the filter bookkeeping uses, on a sorted std::vector.
// The reference keeps map_server and Feature::observations in std::map (msckf_vio.h / feature.hpp:139); keys
// there only ever grow (feature ids, state ids are counters), so inserts are appends, and with a few hundred
// live features per stream the node allocations and pointer chasing of std::map were the largest item of the
// host time per frame (bench.py "host_bookkeeping_us_per_stream_frame").  Iteration order (ascending key) and
// the find / operator[] / erase semantics are those of std::map.
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

namespace cg {

template <class K, class V>
class FlatMap {
  public:
    typedef std::pair<K, V> value_type;
    typedef typename std::vector<value_type>::iterator iterator;
    typedef typename std::vector<value_type>::const_iterator const_iterator;

    iterator begin() { return v_.begin(); }
    iterator end() { return v_.end(); }
    const_iterator begin() const { return v_.begin(); }
    const_iterator end() const { return v_.end(); }
    size_t size() const { return v_.size(); }
    bool empty() const { return v_.empty(); }
    void clear() { v_.clear(); }
    void reserve(size_t n) { v_.reserve(n); }
    // positional access (valid until the next insert / erase)
    value_type &nth(size_t i) { return v_[i]; }
    const value_type &nth(size_t i) const { return v_[i]; }
    size_t index_of(const_iterator it) const { return (size_t)(it - v_.begin()); }
    const K &back_key() const { return v_.back().first; }
    const K &front_key() const { return v_.front().first; }

    iterator find(const K &k) {
        if (!v_.empty() && v_.back().first == k) return v_.end() - 1;       // the newest key is the common query
        iterator it = lower(k);
        return (it != v_.end() && it->first == k) ? it : v_.end();
    }
    const_iterator find(const K &k) const {
        if (!v_.empty() && v_.back().first == k) return v_.end() - 1;
        const_iterator it = std::lower_bound(v_.begin(), v_.end(), k, [](const value_type &a, const K &b) { return a.first < b; });
        return (it != v_.end() && it->first == k) ? it : v_.end();
    }
    V &operator[](const K &k) {
        if (v_.empty() || v_.back().first < k) { v_.emplace_back(k, V()); return v_.back().second; }   // append
        iterator it = lower(k);
        if (it != v_.end() && it->first == k) return it->second;
        return v_.insert(it, value_type(k, V()))->second;
    }
    size_t erase(const K &k) {
        iterator it = find(k);
        if (it == v_.end()) return 0;
        v_.erase(it);
        return 1;
    }
    iterator erase(iterator it) { return v_.erase(it); }
    // erase every key of `keys` (any order, duplicates allowed) in one compaction pass
    void erase_many(std::vector<K> keys) {
        if (keys.empty()) return;
        std::sort(keys.begin(), keys.end());
        size_t w = 0, q = 0;
        for (size_t r = 0; r < v_.size(); ++r) {
            while (q < keys.size() && keys[q] < v_[r].first) ++q;
            if (q < keys.size() && keys[q] == v_[r].first) continue;
            if (w != r) v_[w] = std::move(v_[r]);
            ++w;
        }
        v_.resize(w);
    }

  private:
    iterator lower(const K &k) {
        return std::lower_bound(v_.begin(), v_.end(), k, [](const value_type &a, const K &b) { return a.first < b; });
    }
    std::vector<value_type> v_;
};

}  // namespace cg
