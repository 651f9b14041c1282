// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// The reference's hybrid FT + KNN rank fusion, compiled in place: this TU #includes
//   /root/reference/cpp_src/core/nsselecter/selectiteratorcontainer.cc
// (where `SelectIteratorContainer::MergerRankedImpl`, `IdRank<desc>` and `Merged<desc>` live, :1250-1447) and is built with
// -fno-access-control so that the private nested merger can be driven directly.  Output: oracle/_ref/libref_rank.so; every symbol the
// TU references but this path never calls becomes a trap stub generated from `nm -u` (same recipe as libref_ft.so).
//
// Wrapped reference code:
//   RerankerRRF / RerankerLinear                      core/sorting/reranker.h:11-39
//   MergerRankedImpl::operator()(RRF | Linear, Hnsw)  core/nsselecter/selectiteratorcontainer.cc:1326-1423
//   Merged<desc> + the drain of mergeRanked           selectiteratorcontainer.cc:1280-1303, 1486-1552 (USE_PMR as tools/use_pmr.h decides)
//   RanksHolder::InitRRFPositions                     core/nsselecter/ranks_holder.h:61-76
#include "core/nsselecter/selectiteratorcontainer.cc"

#include "core/nsselecter/ranks_holder.h"

namespace {
using namespace reindexer;
using SIC = SelectIteratorContainer;

struct Args {
	const int32_t* knnIds;
	const float* knnRanks;
	size_t nKnn;
	const int32_t* ftIds;
	const float* ftRanks;
	const size_t* ftPositions;
	size_t nFt;
	int32_t* outIds;
	float* outRanks;
	size_t cap;
};

template <SIC::MergeType mergeType, bool desc, VectorMetric metric, typename RR>
size_t run(const RR& rr, const Args& a) {
	static_assert(sizeof(IdType) == sizeof(int32_t) && sizeof(RankT) == sizeof(float));
	const size_t bufSize = mergeType == SIC::MergeType::Intersection ? std::min(a.nFt, a.nKnn) : a.nFt + a.nKnn;
#ifdef USE_PMR
	std::vector<std::byte> buffer;
	buffer.resize((sizeof(typename Merged<desc>::node_type) + sizeof(typename Merged<desc>::value_type)) * bufSize);
	std::pmr::monotonic_buffer_resource pool(buffer.data(), buffer.size());
	Merged<desc> merged{&pool};
#else
	Merged<desc> merged(bufSize);
#endif
	HnswKnnRawResult knn(a.nKnn);
	for (size_t i = 0; i < a.nKnn; ++i) {
		knn.Ids()[i] = IdType::FromNumber(a.knnIds[i]);
		knn.Dists()[i] = RankT(a.knnRanks[i]);
	}
	SIC::MergerRankedImpl<mergeType, desc, metric> impl{merged, IdSetCRef(reinterpret_cast<const IdType*>(a.ftIds), a.nFt),
														 std::span<const RankT>(reinterpret_cast<const RankT*>(a.ftRanks), a.nFt),
														 std::span<const size_t>(a.ftPositions, a.ftPositions ? a.nFt : 0)};
	impl(rr, knn);
	size_t n = 0;
#ifdef USE_PMR
	for (const auto [id, rank] : merged) {
#else
	std::vector<IdRank<desc>> mergedSorted;
	mergedSorted.assign(merged.begin(), merged.end());
	boost::sort::pdqsort_branchless(mergedSorted.begin(), mergedSorted.end());
	for (const auto [id, rank] : mergedSorted) {
#endif
		if (n < a.cap) {
			a.outIds[n] = id.ToNumber();
			a.outRanks[n] = rank.Value();
		}
		++n;
	}
	return n;
}

template <SIC::MergeType mergeType, bool desc, typename RR>
size_t byMetric(int metric, const RR& rr, const Args& a) {
	switch (metric) {
		case 0:
			return run<mergeType, desc, VectorMetric::L2>(rr, a);
		case 1:
			return run<mergeType, desc, VectorMetric::InnerProduct>(rr, a);
		default:
			return run<mergeType, desc, VectorMetric::Cosine>(rr, a);
	}
}
template <typename RR>
size_t dispatch(int unionMerge, int desc, int metric, const RR& rr, const Args& a) {
	if (unionMerge) {
		return desc ? byMetric<SIC::MergeType::Union, true>(metric, rr, a) : byMetric<SIC::MergeType::Union, false>(metric, rr, a);
	}
	return desc ? byMetric<SIC::MergeType::Intersection, true>(metric, rr, a) : byMetric<SIC::MergeType::Intersection, false>(metric, rr, a);
}
}  // namespace

extern "C" {

// 1 = std::pmr::set keyed by (rank, id) [the default gcc build], 0 = hash set keyed by id + pdqsort
int ref_rank_uses_pmr() {
#ifdef USE_PMR
	return 1;
#else
	return 0;
#endif
}

// ranks in FT result order (best first) -> 1-based RRF positions
void ref_rank_init_rrf_positions(const float* ranks, size_t n, size_t* out) {
	RanksHolder h;
	h_vector<RankT, 128> r;
	r.reserve(n);
	for (size_t i = 0; i < n; ++i) r.push_back(RankT(ranks[i]));
	h.Add(std::move(r));
	h.InitRRFPositions();
	const auto pos = h.GetPositionsSpan();
	for (size_t i = 0; i < pos.size(); ++i) out[i] = pos[i];
}

// returns the merged size (may exceed cap; only cap entries are written), -1 on exception
long ref_rank_merge_rrf(double rankConst, int unionMerge, int desc, int metric, const int32_t* knnIds, const float* knnRanks, size_t nKnn,
						const int32_t* ftIds, const float* ftRanks, const size_t* ftPositions, size_t nFt, int32_t* outIds, float* outRanks, size_t cap) {
	try {
		return long(dispatch(unionMerge, desc, metric, RerankerRRF(rankConst), Args{knnIds, knnRanks, nKnn, ftIds, ftRanks, ftPositions, nFt, outIds, outRanks, cap}));
	} catch (...) {
		return -1;
	}
}
long ref_rank_merge_linear(double kKnn, double knnDefault, double kFt, double ftDefault, double c, int unionMerge, int desc, int metric,
						   const int32_t* knnIds, const float* knnRanks, size_t nKnn, const int32_t* ftIds, const float* ftRanks, size_t nFt,
						   int32_t* outIds, float* outRanks, size_t cap) {
	try {
		return long(dispatch(unionMerge, desc, metric, RerankerLinear(kKnn, knnDefault, kFt, ftDefault, c),
							 Args{knnIds, knnRanks, nKnn, ftIds, ftRanks, nullptr, nFt, outIds, outRanks, cap}));
	} catch (...) {
		return -1;
	}
}
}
