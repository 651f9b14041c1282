// Mirror of the engine-facing half of HnswIndexBase<Map> (cpp_src/core/index/float_vector/hnsw_index.cc:159-288):
// search() (query normalisation for cosine, k / radius unpacking), select() (drain best-first, rank sign, rowId
// extraction, equal-distance runs sorted by id, array dedupe, removeOverK) and selectRaw().  Templated on the Map
// exactly like the reference, so GpuBruteforceMap / GpuHnswMap slot in where BruteforceSearch / HierarchicalNSW do.
#pragma once

#include <algorithm>
#include <optional>
#include <stdexcept>
#include <unordered_set>
#include <vector>

#include "gpu_bruteforce_map.h"
#include "rx_types.h"

namespace rxgpu::host {

// core/query/knn_search_params.h:147-192 (brute-force / hnsw flavours folded into one POD)
struct KnnSearchParams {
	std::optional<size_t> k;
	std::optional<float> radius;
	size_t ef = 0;   // HNSW only; planner default ef = k (knn_search_params.cc:25-38)

	// knn_search_params.cc:176-210
	void Validate(bool isHnsw) const {
		if (k && *k == 0) throw std::invalid_argument("KNN limit should not be 0");
		if (!isHnsw && !k && !radius) throw std::invalid_argument("K and Radius params can not be empty both");
		if (isHnsw && k && ef < *k) throw std::invalid_argument("Ef should not be less than k in hnsw query");
	}
};

struct KnnSelectResult {
	std::vector<int32_t> ids;   // IdSetPlain contents (rowIds), best first
	std::vector<float> ranks;   // RanksHolder contents: L2 as is, IP / cosine sign-flipped
};

// float_vector_index.cc:319-333 checkForSelect
inline void CheckForSelect(ConstFloatVectorView key, size_t indexDim) {
	if (key.IsEmpty()) throw std::invalid_argument("Attempt to search knn by empty float vector");
	if (key.Dimension() != indexDim) {
		throw std::invalid_argument("Attempt to search vector of dimension " + std::to_string(key.Dimension()) +
									" in a float vector index of dimension " + std::to_string(indexDim));
	}
}

// hnsw_index.cc:159-191
template <typename Map>
SearchResultQueue KnnSearch(const Map& map, ConstFloatVectorView key, const KnnSearchParams& params,
							std::optional<float> indexDefaultRadius = std::nullopt) {
	std::vector<float> normalized;
	std::optional<float> normL2;
	const float* keyData = key.Data();
	if (map.Metric() == VectorMetric::Cosine) {
		normalized.resize(key.Dimension());
		normL2 = 1.f / NormalizeCopyVector(key.Data(), int32_t(key.Dimension()), normalized.data());
		keyData = normalized.data();
	}
	const std::optional<float> radius = params.radius ? params.radius : indexDefaultRadius;
	if (radius) {
		return map.SearchRange(keyData, normL2, map.Metric() == VectorMetric::L2 ? *radius : -*radius, params.ef);
	}
	return map.SearchKnn(keyData, normL2, *params.k, params.ef);
}

// float_vector_index.h:140-160
inline void RemoveDuplicateRowId(std::vector<int32_t>& ids, std::vector<float>& ranks) {
	std::unordered_set<int32_t> added;
	added.reserve(ids.size());
	size_t to = 0;
	for (size_t from = 0; from < ids.size(); ++from) {
		if (added.insert(ids[from]).second) {
			ids[to] = ids[from];
			ranks[to] = ranks[from];
			++to;
		}
	}
	ids.resize(to);
	ranks.resize(to);
}

// hnsw_index.cc:231-288 (needSort = KnnCtx::NeedSort(): no explicit ORDER BY; isArray = Opts().IsArray())
template <typename Map>
KnnSelectResult KnnSelect(const Map& map, ConstFloatVectorView key, const KnnSearchParams& params, bool needSort, bool isArray,
						  std::optional<float> indexDefaultRadius = std::nullopt) {
	CheckForSelect(key, map.Dim());
	auto knnRes = KnnSearch(map, key, params, indexDefaultRadius);
	KnnSelectResult out;
	const size_t n = knnRes.size();
	if (n) {
		out.ids.resize(n);
		out.ranks.resize(n);
		const bool l2 = map.Metric() == VectorMetric::L2;
		for (size_t i = n; !knnRes.empty(); knnRes.pop()) {
			--i;
			out.ranks[i] = l2 ? knnRes.top().first : -knnRes.top().first;
			out.ids[i] = FloatVectorId::FromNumber(knnRes.top().second).RowId();
		}
		if (needSort) {
			size_t runStart = 0;
			for (size_t i = 1; i <= n; ++i) {
				if (i == n || out.ranks[i] != out.ranks[runStart]) {
					std::sort(out.ids.begin() + runStart, out.ids.begin() + i);
					runStart = i;
				}
			}
		}
		if (isArray) RemoveDuplicateRowId(out.ids, out.ranks);
		// removeOverK (:193-203)
		if (params.k && (params.radius || indexDefaultRadius) && out.ids.size() > *params.k) {
			out.ids.resize(*params.k);
			out.ranks.resize(*params.k);
		}
	}
	return out;
}

// hnsw_index.cc:205-229 (hybrid queries: no equal-rank id sort)
template <typename Map>
KnnSelectResult KnnSelectRaw(const Map& map, ConstFloatVectorView key, const KnnSearchParams& params, bool isArray,
							 std::optional<float> indexDefaultRadius = std::nullopt) {
	return KnnSelect(map, key, params, /*needSort*/ false, isArray, indexDefaultRadius);
}

// hnsw_index.cc:290-351: beginStreaming / continueStreaming (HNSW maps only; the brute-force specialisations throw errQueryExec,
// hnsw_index.cc:353-361).  The session keeps the normalised cosine query alive like HnswStreamingSessionImpl::queryStorage.
template <typename Session>
struct KnnStreamingSession {
	std::vector<float> queryStorage;
	Session session;
};
struct KnnStreamingBatch {   // core/index/float_vector/knn_streaming.h: best first
	std::vector<int32_t> ids;
	std::vector<float> ranks;
	bool exhausted = false;
};
template <typename Map>
auto KnnBeginStreaming(const Map& map, ConstFloatVectorView key, size_t ef) {
	using Session = decltype(map.BeginStreamingSearch(nullptr, std::nullopt, {}));
	KnnStreamingSession<Session> s;
	const float* keyData = key.Data();
	std::optional<float> normL2;
	if (map.Metric() == VectorMetric::Cosine) {
		s.queryStorage.resize(key.Dimension());
		normL2 = 1.f / NormalizeCopyVector(key.Data(), int32_t(key.Dimension()), s.queryStorage.data());
		keyData = s.queryStorage.data();
	}
	s.session = map.BeginStreamingSearch(keyData, normL2, {ef});
	return s;
}
template <typename Map, typename Session>
void KnnContinueStreaming(const Map& map, KnnStreamingSession<Session>& session, size_t batchSize, KnnStreamingBatch& out) {
	auto hnswBatch = map.ContinueStreamingSearch(session.session, batchSize);
	auto& results = hnswBatch.results;
	out.exhausted = hnswBatch.exhausted;
	const size_t n = results.size();
	out.ids.resize(n);
	out.ranks.resize(n);
	for (size_t i = n; !results.empty(); results.pop()) {
		--i;
		// IP and cosine are sorted in reverse order inside the engine and carry the opposite sign (hnsw_index.cc:336-345)
		out.ranks[i] = map.Metric() == VectorMetric::L2 ? results.top().first : -results.top().first;
		out.ids[i] = FloatVectorId::FromNumber(results.top().second).RowId();
	}
}

}  // namespace rxgpu::host
