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
#include <unordered_set>
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
// IdRank<desc>::operator< (selectiteratorcontainer.cc:1260-1281): by rank (descending when desc), ties by ascending id.  Sorted through one
// 64-bit key per entry (order-preserving image of the float in the high word, id in the low word): same order, no comparator calls.
inline void Finish(std::vector<IdRank>& merged, bool desc, HybridResult& out) {
	struct Key {
		uint64_t key;
		uint32_t idx;
	};
	std::vector<Key> keys(merged.size());
	for (size_t i = 0; i < merged.size(); ++i) {
		uint32_t u;
		const float r = merged[i].rank + 0.0f;   // -0 and +0 compare equal
		std::memcpy(&u, &r, sizeof(u));
		u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending float order == ascending unsigned order
		if (desc) u = ~u;
		keys[i] = Key{(uint64_t(u) << 32) | uint32_t(merged[i].id), uint32_t(i)};   // ids are non-negative (IdType) and unique
	}
	std::sort(keys.begin(), keys.end(), [](const Key& l, const Key& r) { return l.key < r.key; });
	out.ids.resize(keys.size());
	out.ranks.resize(keys.size());
	for (size_t i = 0; i < keys.size(); ++i) {
		out.ids[i] = merged[keys[i].idx].id;
		out.ranks[i] = merged[keys[i].idx].rank;
	}
}
}  // namespace detail

// RRF: knnIds/knnRanks best-first as returned by KnnSelectRaw (L2: ascending distance; IP / cosine: descending similarity);
// ftIds ascending with ftPositions from InitRRFPositions over the FT ranks in FT result order.
inline HybridResult MergeRankedRRF(const RerankerRRF& rr, HybridMergeType type, bool desc, VectorMetric metric, const std::vector<int32_t>& knnIds,
								   const std::vector<float>& knnRanks, const std::vector<int32_t>& ftIds, const std::vector<size_t>& ftPositions) {
	std::vector<detail::IdRank> merged;
	// Merged<desc> is keyed by id: the first emplace of an id wins.  Only the (few) KNN ids can repeat: ftIds are unique, and an FT id that
	// also came through the KNN list is marked in ftAdded — so the FT tail needs no hashing.
	std::unordered_set<int32_t> seen;
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
				if (seen.insert(id).second) merged.push_back({id, rr.Calculate(knnPos, ftPositions[n])});
				if (type == HybridMergeType::Union) ftAdded[n] = true;
			} else if (type == HybridMergeType::Union) {
				if (seen.insert(id).second) merged.push_back({id, rr.CalculateSingle(knnPos)});
			}
		}
	}
	if (type == HybridMergeType::Union) {
		for (size_t i = 0; i < ftIds.size(); ++i) {
			if (!ftAdded[i]) merged.push_back({ftIds[i], rr.CalculateSingle(ftPositions[i])});
		}
	}
	HybridResult out;
	detail::Finish(merged, desc, out);
	return out;
}

inline HybridResult MergeRankedLinear(const RerankerLinear& rr, HybridMergeType type, bool desc, const std::vector<int32_t>& knnIds,
									  const std::vector<float>& knnRanks, const std::vector<int32_t>& ftIds, const std::vector<float>& ftRanks) {
	std::vector<detail::IdRank> merged;
	std::unordered_set<int32_t> seen;
	std::vector<bool> ftAdded(type == HybridMergeType::Union ? ftIds.size() : 0, false);
	for (size_t i = 0; i < knnIds.size(); ++i) {
		const int32_t id = knnIds[i];
		auto it = std::lower_bound(ftIds.begin(), ftIds.end(), id);
		if (it != ftIds.end() && *it == id) {
			const size_t n = size_t(it - ftIds.begin());
			if (seen.insert(id).second) merged.push_back({id, rr.Calculate(double(knnRanks[i]), ftRanks[n])});
			if (type == HybridMergeType::Union) ftAdded[n] = true;
		} else if (type == HybridMergeType::Union) {
			if (seen.insert(id).second) merged.push_back({id, rr.CalculateJustKnn(double(knnRanks[i]))});
		}
	}
	if (type == HybridMergeType::Union) {
		for (size_t i = 0; i < ftIds.size(); ++i) {
			if (!ftAdded[i]) merged.push_back({ftIds[i], rr.CalculateJustFt(ftRanks[i])});   // unique ids, not in the KNN list: no hashing needed
		}
	}
	HybridResult out;
	detail::Finish(merged, desc, out);
	return out;
}

}  // namespace rxgpu::host
