// Hybrid FT + KNN rank fusion — the step right after both GPU engines in `SELECT ... WHERE ft = '...' AND/OR KNN(...) ORDER BY RRF()`
// (BASELINE configs[4]).  Mirrors, on the host (k + |FT| <= ~20k rows, scalar math — SURVEY §8f row 1):
//   RanksHolder::InitRRFPositions             cpp_src/core/nsselecter/ranks_holder.h:61-76
//   RerankerRRF / RerankerLinear              cpp_src/core/sorting/reranker.h:11-39
//   MergerRankedImpl::operator() + mergeRanked cpp_src/core/nsselecter/selectiteratorcontainer.cc:1343-1423, 1454-1559
// Inputs are exactly what the two engines hand back: KnnSelectRaw (ids best-first + ranks, knn_select.h) and the FT id set
// (ascending ids) with its ranks / RRF positions.
#pragma once

#include <algorithm>
#include <cstring>
#include <cstdint>
#include <vector>

#include "rx_types.h"

namespace rxgpu::host {

enum class HybridMergeType { Intersection, Union };   // second ranked condition joined with AND / OR (selectiteratorcontainer.cc:1473-1480)

struct RerankerRRF {
	double rankConst = 60.0;   // hybrid.md: rank_const default
	float Calculate(size_t posKnn, size_t posFt) const noexcept { return float(1.0 / (rankConst + double(posKnn)) + 1.0 / (rankConst + double(posFt))); }
	float CalculateSingle(size_t pos) const noexcept { return float(1.0 / (rankConst + double(pos))); }
};
struct RerankerLinear {
	double kKnn = 1.0, knnDefault = 0.0, kFt = 1.0, ftDefault = 0.0, c = 0.0;
	float Calculate(double rankKnn, float rankFt) const noexcept { return float(kKnn * rankKnn + kFt * double(rankFt) + c); }
	float CalculateJustKnn(double rankKnn) const noexcept { return float(kKnn * rankKnn + kFt * ftDefault + c); }
	float CalculateJustFt(float rankFt) const noexcept { return float(kKnn * knnDefault + kFt * double(rankFt) + c); }
};

// positions_[i] = 1-based position of the first element of i's run of equal ranks (ranks are sorted best-first: descending)
inline std::vector<size_t> InitRRFPositions(const std::vector<float>& ranks) {
	std::vector<size_t> pos(ranks.size());
	if (ranks.empty()) return pos;
	size_t p = 1;
	float last = ranks.front();
	for (size_t i = 0; i < ranks.size(); ++i) {
		if (ranks[i] < last) {
			last = ranks[i];
			p = i + 1;
		}
		pos[i] = p;
	}
	return pos;
}

struct HybridResult {
	std::vector<int32_t> ids;
	std::vector<float> ranks;
};

namespace detail {
struct IdRank {
	int32_t id;
	float rank;
};
// order-preserving image of a float in an unsigned word (ascending float order == ascending unsigned order; -0 and +0 share one image)
inline uint32_t SortableBits(float v) noexcept {
	uint32_t u;
	const float r = v + 0.0f;
	std::memcpy(&u, &r, sizeof(u));
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// Stable LSD radix sort of (key, value) pairs by key: 11-bit digits, digits on which all keys agree are skipped (ranks are a few hundred
// distinct values, ids fit 23 bits — 3-4 passes of ~n steps each instead of n log n comparator calls).
template <typename K>
inline void RadixSortPairs(std::vector<K>& keys, std::vector<uint32_t>& vals) {
	const size_t n = keys.size();
	if (n < 2) return;
	constexpr int kBits = 11, kBuckets = 1 << kBits, kPasses = (int(sizeof(K)) * 8 + kBits - 1) / kBits;
	std::vector<uint32_t> hist(size_t(kPasses) * kBuckets, 0u);
	for (size_t i = 0; i < n; ++i) {
		const K k = keys[i];
		for (int p = 0; p < kPasses; ++p) ++hist[size_t(p) * kBuckets + ((k >> (p * kBits)) & (kBuckets - 1))];
	}
	std::vector<K> keys2(n);
	std::vector<uint32_t> vals2(n);
	K* ka = keys.data();
	K* kb = keys2.data();
	uint32_t* va = vals.data();
	uint32_t* vb = vals2.data();
	for (int p = 0; p < kPasses; ++p) {
		uint32_t* h = hist.data() + size_t(p) * kBuckets;
		if (h[(ka[0] >> (p * kBits)) & (kBuckets - 1)] == n) continue;   // every key has this digit
		uint32_t sum = 0;
		for (int b = 0; b < kBuckets; ++b) {
			const uint32_t c = h[b];
			h[b] = sum;
			sum += c;
		}
		for (size_t i = 0; i < n; ++i) {
			const uint32_t dst = h[(ka[i] >> (p * kBits)) & (kBuckets - 1)]++;
			kb[dst] = ka[i];
			vb[dst] = va[i];
		}
		std::swap(ka, kb);
		std::swap(va, vb);
	}
	if (ka != keys.data()) {
		keys.swap(keys2);
		vals.swap(vals2);
	}
}
// IdRank<desc>::operator< (selectiteratorcontainer.cc:1258-1278): desc == false: rank ascending, ties by ASCENDING id; desc == true (the
// normal RRF / rank() ordering): rank descending, ties by DESCENDING id.  The container mirrored is the default (gcc) build's
// Merged<desc> = std::pmr::set<IdRank<desc>> (tools/use_pmr.h, selectiteratorcontainer.cc:1280-1283): its key is the (rank, id) PAIR, so a
// row id that reaches the merger twice (two vectors of one array row in the KNN list) stays twice unless both ranks are equal too.
// merged[0, tailStart) are the (few) entries that came through the KNN list, in no particular order; merged[tailStart, n) is the FT-only
// tail in ascending id order, disjoint from the head (an FT id that is in the KNN list is marked in ftAdded).  A STABLE sort of the tail
// by rank alone — fed in reverse for desc, so equal ranks come out in descending id order — yields its final order (32-bit radix keys,
// order-preserving image of the float); the head is sorted by comparison and stripped of exact (rank, id) repeats; one linear merge by
// (rank key, id) finishes.
inline void Finish(std::vector<IdRank>& merged, size_t tailStart, bool desc, HybridResult& out) {
	const size_t n = merged.size();
	auto rankKey = [desc](float r) noexcept {
		const uint32_t u = SortableBits(r);
		return desc ? ~u : u;
	};
	auto idBefore = [desc](int32_t l, int32_t r) noexcept { return desc ? l > r : l < r; };
	const size_t nTail = n - tailStart;
	std::vector<uint32_t> tkey(nTail), tidx(nTail);
	for (size_t j = 0; j < nTail; ++j) {
		const size_t i = desc ? n - 1 - j : tailStart + j;
		tkey[j] = rankKey(merged[i].rank);
		tidx[j] = uint32_t(i);
	}
	RadixSortPairs(tkey, tidx);
	std::vector<uint64_t> hkey(tailStart);
	for (size_t i = 0; i < tailStart; ++i) hkey[i] = (uint64_t(rankKey(merged[i].rank)) << 32) | (uint64_t(i) & 0xFFFFFFFFull);
	std::sort(hkey.begin(), hkey.end(), [&](uint64_t l, uint64_t r) {
		const uint32_t lk = uint32_t(l >> 32), rk = uint32_t(r >> 32);
		return lk != rk ? lk < rk : idBefore(merged[uint32_t(l)].id, merged[uint32_t(r)].id);
	});
	hkey.erase(std::unique(hkey.begin(), hkey.end(),
						   [&](uint64_t l, uint64_t r) { return uint32_t(l >> 32) == uint32_t(r >> 32) && merged[uint32_t(l)].id == merged[uint32_t(r)].id; }),
			   hkey.end());
	const size_t total = hkey.size() + nTail;
	out.ids.resize(total);
	out.ranks.resize(total);
	size_t h = 0, t = 0, o = 0;
	while (h < hkey.size() || t < tkey.size()) {
		bool takeHead;
		if (h == hkey.size()) {
			takeHead = false;
		} else if (t == tkey.size()) {
			takeHead = true;
		} else {
			const uint32_t hk = uint32_t(hkey[h] >> 32);
			takeHead = hk != tkey[t] ? hk < tkey[t] : idBefore(merged[uint32_t(hkey[h])].id, merged[tidx[t]].id);
		}
		const IdRank& e = takeHead ? merged[uint32_t(hkey[h++])] : merged[tidx[t++]];
		out.ids[o] = e.id;
		out.ranks[o] = e.rank;
		++o;
	}
}
}  // namespace detail

// What the FT engine hands back is in FT result order (best rank first; ft::Merger / GpuFtMerger with sortByRank).  The fusion wants the id
// set ascending (ftIds_) with, for RRF, each document's position in the rank order (RanksHolder::InitRRFPositions).  One pass each: the
// positions come straight from the given order when it is rank-descending (else a stable rank sort restores it), the id order from a
// radix sort of (id, index).
struct FtById {
	std::vector<int32_t> ids;        // ascending
	std::vector<float> ranks;        // aligned with ids
	std::vector<size_t> positions;   // aligned with ids (RRF)
};
inline FtById PrepareFtById(const std::vector<int32_t>& ftIdsFtOrder, const std::vector<float>& ftRanksFtOrder) {
	const size_t n = ftIdsFtOrder.size();
	FtById out;
	std::vector<size_t> posFtOrder;
	if (std::is_sorted(ftRanksFtOrder.begin(), ftRanksFtOrder.end(), [](float a, float b) { return a > b; })) {
		posFtOrder = InitRRFPositions(ftRanksFtOrder);
	} else {
		std::vector<uint64_t> keys(n);
		std::vector<uint32_t> idx(n);
		for (size_t i = 0; i < n; ++i) {
			keys[i] = uint64_t(~detail::SortableBits(ftRanksFtOrder[i])) << 32 | uint32_t(i);
			idx[i] = uint32_t(i);
		}
		detail::RadixSortPairs(keys, idx);
		std::vector<float> sorted(n);
		for (size_t i = 0; i < n; ++i) sorted[i] = ftRanksFtOrder[idx[i]];
		const auto posSorted = InitRRFPositions(sorted);
		posFtOrder.resize(n);
		for (size_t i = 0; i < n; ++i) posFtOrder[idx[i]] = posSorted[i];
	}
	std::vector<uint32_t> keys(n), idx(n);
	for (size_t i = 0; i < n; ++i) {
		keys[i] = uint32_t(ftIdsFtOrder[i]);
		idx[i] = uint32_t(i);
	}
	detail::RadixSortPairs(keys, idx);
	out.ids.resize(n);
	out.ranks.resize(n);
	out.positions.resize(n);
	for (size_t i = 0; i < n; ++i) {
		out.ids[i] = ftIdsFtOrder[idx[i]];
		out.ranks[i] = ftRanksFtOrder[idx[i]];
		out.positions[i] = posFtOrder[idx[i]];
	}
	return out;
}

// RRF: knnIds/knnRanks best-first as returned by KnnSelectRaw (L2: ascending distance; IP / cosine: descending similarity);
// ftIds ascending with ftPositions from InitRRFPositions over the FT ranks in FT result order.
inline HybridResult MergeRankedRRF(const RerankerRRF& rr, HybridMergeType type, bool desc, VectorMetric metric, const std::vector<int32_t>& knnIds,
								   const std::vector<float>& knnRanks, const std::vector<int32_t>& ftIds, const std::vector<size_t>& ftPositions) {
	std::vector<detail::IdRank> merged;
	// Merged<desc> (std::pmr::set) is keyed by the (rank, id) pair: only the (few) KNN ids can repeat, and such repeats are resolved by
	// detail::Finish; ftIds are unique, and an FT id that also came through the KNN list is marked in ftAdded.
	merged.reserve(knnIds.size() + (type == HybridMergeType::Union ? ftIds.size() : 0));
	std::vector<bool> ftAdded(type == HybridMergeType::Union ? ftIds.size() : 0, false);
	if (!knnIds.empty()) {
		float last = knnRanks.front();
		size_t knnPos = 1;
		for (size_t i = 0; i < knnIds.size(); ++i) {
			if (metric == VectorMetric::L2 ? last < knnRanks[i] : last > knnRanks[i]) {
				last = knnRanks[i];
				knnPos = i + 1;
			}
			const int32_t id = knnIds[i];
			auto it = std::lower_bound(ftIds.begin(), ftIds.end(), id);
			if (it != ftIds.end() && *it == id) {
				const size_t n = size_t(it - ftIds.begin());
				merged.push_back({id, rr.Calculate(knnPos, ftPositions[n])});
				if (type == HybridMergeType::Union) ftAdded[n] = true;
			} else if (type == HybridMergeType::Union) {
				merged.push_back({id, rr.CalculateSingle(knnPos)});
			}
		}
	}
	const size_t tailStart = merged.size();
	if (type == HybridMergeType::Union) {
		for (size_t i = 0; i < ftIds.size(); ++i) {
			if (!ftAdded[i]) merged.push_back({ftIds[i], rr.CalculateSingle(ftPositions[i])});
		}
	}
	HybridResult out;
	detail::Finish(merged, tailStart, desc, out);
	return out;
}

inline HybridResult MergeRankedLinear(const RerankerLinear& rr, HybridMergeType type, bool desc, const std::vector<int32_t>& knnIds,
									  const std::vector<float>& knnRanks, const std::vector<int32_t>& ftIds, const std::vector<float>& ftRanks) {
	std::vector<detail::IdRank> merged;
	std::vector<bool> ftAdded(type == HybridMergeType::Union ? ftIds.size() : 0, false);
	for (size_t i = 0; i < knnIds.size(); ++i) {
		const int32_t id = knnIds[i];
		auto it = std::lower_bound(ftIds.begin(), ftIds.end(), id);
		if (it != ftIds.end() && *it == id) {
			const size_t n = size_t(it - ftIds.begin());
			merged.push_back({id, rr.Calculate(double(knnRanks[i]), ftRanks[n])});
			if (type == HybridMergeType::Union) ftAdded[n] = true;
		} else if (type == HybridMergeType::Union) {
			merged.push_back({id, rr.CalculateJustKnn(double(knnRanks[i]))});
		}
	}
	const size_t tailStart = merged.size();
	if (type == HybridMergeType::Union) {
		for (size_t i = 0; i < ftIds.size(); ++i) {
			if (!ftAdded[i]) merged.push_back({ftIds[i], rr.CalculateJustFt(ftRanks[i])});   // unique ids, not in the KNN list: no hashing needed
		}
	}
	HybridResult out;
	detail::Finish(merged, tailStart, desc, out);
	return out;
}

}  // namespace rxgpu::host
