// ORACLE (test infrastructure, CPU only) -- not part of the shipped product path.
//
// RHMap<K,V>: restatement of the *observable iteration order* of the reference's vendored
// robin_hood::unordered_flat_map 3.11.1 for integral keys. The reference lets that order leak into
// results through tie-breaks (call_variants.cpp:477-501, :837-844; Partition.cpp:59-66;
// separate_reads.cpp:1086-1099), so the oracle has to reproduce it.
// Follows: hash_int robin_hood.h:749-760; keyToIdx :1349-1361; insertKeyPrepareEmptySpot :2330-2380;
// shiftUp :1377-1397; insert_move :1451-1493; try_increase_info :2383-2411; increase_size :2413-2443;
// rehashPowerOfTwo :2203-2237; initData :2305-2325; sizes :945-949,2123-2142; iteration = ascending slot.
// Pinned by tests/golden/robin_hood_order.json (generated from the real header by oracle/gen_goldens.py).
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <stdexcept>

namespace hso {

template <class K, class V>
class RHMap {
public:
    struct Slot { K first; V second; };

    RHMap() = default;

    size_t size() const { return num_; }
    bool empty() const { return num_ == 0; }

    // returns slot index or -1
    long find_idx(K key) const {
        if (mask_ == 0) return -1;
        size_t idx; uint32_t info;
        key_to_idx(key, idx, info);
        while (info < info_[idx]) { idx++; info += inc_; }
        while (info == info_[idx]) {
            if (slots_[idx].first == key) return (long)idx;
            idx++; info += inc_;
        }
        return -1;
    }
    bool contains(K key) const { return find_idx(key) >= 0; }

    V& operator[](K key) {
        size_t idx = insert_key(key);
        return slots_[idx].second;
    }
    const V& at(K key) const {
        long i = find_idx(key);
        if (i < 0) throw std::out_of_range("RHMap::at");
        return slots_[(size_t)i].second;
    }

    // iteration in the reference's order: ascending slot index over occupied slots
    template <class F> void for_each(F f) const {
        for (size_t i = 0; i < info_.size(); i++) if (info_[i] != 0) f(slots_[i].first, slots_[i].second);
    }
    std::vector<Slot> items() const {
        std::vector<Slot> out;
        for (size_t i = 0; i < info_.size(); i++) if (info_[i] != 0) out.push_back(slots_[i]);
        return out;
    }

private:
    static uint64_t hash_int(uint64_t x) {
        x ^= x >> 33U; x *= UINT64_C(0xff51afd7ed558ccd); x ^= x >> 33U; return x;
    }
    static size_t calc_max(size_t n) { return n * 80 / 100; }
    static size_t with_buffer(size_t n) { size_t m = calc_max(n); return n + (m < 0xFF ? m : 0xFF); }

    void key_to_idx(K key, size_t& idx, uint32_t& info) const {
        // integral keys are widened exactly as static_cast<uint64_t>(obj) does (sign-extending)
        uint64_t h = hash_int((uint64_t)(int64_t)key);
        h *= mult_; h ^= h >> 33U;
        info = inc_ + (uint32_t)((h & 31U) >> shift_);
        idx = (size_t)(h >> 5U) & mask_;
    }
    void init_data(size_t n) {
        num_ = 0; mask_ = n - 1; max_allowed_ = calc_max(n);
        size_t nb = with_buffer(n);
        info_.assign(nb + 8, 0);      // + sentinel/padding region
        info_[nb] = 0;                // sentinel is not an element; keep it 0 for iteration
        sentinel_ = nb;
        slots_.assign(nb + 8, Slot{});
        inc_ = 32; shift_ = 0;
    }
    // the real table keeps a non-zero sentinel byte at [nb]; probes can never run past it because the
    // buffer is as large as the maximum displacement. We emulate it with an explicit bound check.
    uint8_t info_at(size_t i) const { return i == sentinel_ ? 1 : info_[i]; }

    bool try_increase_info() {
        if (inc_ <= 2) return false;
        inc_ >>= 1; shift_++;
        for (size_t i = 0; i < sentinel_; i++) info_[i] = (uint8_t)((info_[i] >> 1) & 0x7f);
        max_allowed_ = calc_max(mask_ + 1);
        return true;
    }
    void shift_up(size_t start, size_t ins) {
        for (size_t i = start; i != ins; i--) slots_[i] = slots_[i - 1];
        for (size_t i = start; i != ins; i--) {
            info_[i] = (uint8_t)(info_[i - 1] + inc_);
            if ((uint32_t)info_[i] + inc_ > 0xFF) max_allowed_ = 0;
        }
    }
    void insert_move(const Slot& kv) {
        if (max_allowed_ == 0 && !try_increase_info()) throw std::overflow_error("RHMap overflow");
        size_t idx; uint32_t info;
        key_to_idx(kv.first, idx, info);
        while (info <= info_at(idx)) { idx++; info += inc_; }
        size_t ins = idx; uint8_t ins_info = (uint8_t)info;
        if ((uint32_t)ins_info + inc_ > 0xFF) max_allowed_ = 0;
        while (info_at(idx) != 0) idx++;
        if (idx != ins) shift_up(idx, ins);
        slots_[ins] = kv; info_[ins] = ins_info; num_++;
    }
    void rehash(size_t n) {
        std::vector<uint8_t> old_info; old_info.swap(info_);
        std::vector<Slot> old_slots; old_slots.swap(slots_);
        size_t old_n = sentinel_;
        init_data(n);
        for (size_t i = 0; i < old_n; i++) if (old_info[i] != 0) insert_move(old_slots[i]);
    }
    void increase_size() {
        if (mask_ == 0) { init_data(8); return; }
        size_t mx = calc_max(mask_ + 1);
        if (num_ < mx && try_increase_info()) return;
        mult_ += UINT64_C(0xc4ceb9fe1a85ec54);
        if (num_ * 2 < calc_max(mask_ + 1)) rehash(mask_ + 1);
        else rehash((mask_ + 1) * 2);
    }
    size_t insert_key(K key) {
        for (int attempt = 0; attempt < 256; attempt++) {
            if (mask_ != 0) {
                size_t idx; uint32_t info;
                key_to_idx(key, idx, info);
                while (info < info_at(idx)) { idx++; info += inc_; }
                while (info == info_at(idx)) {
                    if (slots_[idx].first == key) return idx;
                    idx++; info += inc_;
                }
                if (num_ >= max_allowed_) { increase_size(); continue; }
                size_t ins = idx; uint32_t ins_info = info;
                if (ins_info + inc_ > 0xFF) max_allowed_ = 0;
                while (info_at(idx) != 0) idx++;
                if (idx != ins) shift_up(idx, ins);
                info_[ins] = (uint8_t)ins_info; num_++;
                slots_[ins].first = key; slots_[ins].second = V{};
                return ins;
            }
            increase_size();
        }
        throw std::overflow_error("RHMap overflow");
    }

    uint64_t mult_ = UINT64_C(0xc4ceb9fe1a85ec53);
    size_t num_ = 0, mask_ = 0, max_allowed_ = 0, sentinel_ = 0;
    uint32_t inc_ = 32, shift_ = 0;
    std::vector<uint8_t> info_;
    std::vector<Slot> slots_;
};

}  // namespace hso
