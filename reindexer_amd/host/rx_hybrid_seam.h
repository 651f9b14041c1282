// In-tree adapter (RXGPU_IN_TREE only) for the hybrid rank fusion: what SelectIteratorContainer::mergeRanked
// (cpp_src/core/nsselecter/selectiteratorcontainer.cc:1454-1559) does with its two ranked conditions — MergerRankedImpl over the KNN raw result
// and the full-text ids / ranks (:1343-1447), then the Merged<desc> order (:1258-1283, :1536-1546) — handed to the device fusion
// (rxgpu_hybrid_fuse, hybrid_fuse.hip) with the reference's own types on both sides.  integration/patches/0005-hybrid-fusion-on-gpu.patch
// calls FuseRankedOnGpu from mergeRanked; tests/test_seam_compile.py compiles the patched translation unit.
//
// The fully resident form — both halves computed on the device and nothing in between leaving HBM — is hybrid_query.h's HybridQueryResident;
// it needs the query (FT DSL + KNN key), which mergeRanked no longer sees: the planner binds it where it still holds both query entries
// (INTEGRATION.md §3g).
#pragma once
#if !defined(RXGPU_IN_TREE)
#error "rx_hybrid_seam.h is for the build inside cpp_src (define RXGPU_IN_TREE)"
#endif

#include <cmath>
#include <cstdlib>
#include <span>
#include <variant>
#include <vector>

#include "core/enums.h"
#include "core/id_type.h"
#include "core/index/float_vector/float_vector_id.h"
#include "core/index/float_vector/knn_raw_result.h"
#include "core/rank_t.h"
#include "core/sorting/reranker.h"
#include "device_list.h"
#include "rxgpu.h"

namespace rxgpu::host {

// RX_GPU_HYBRID=<device> routes the rank fusion of hybrid queries to the MI355X (unset / empty: the CPU hash merge)
inline int HybridDeviceFromEnv() noexcept {
	const std::vector<int> d = GpuDevicesFromEnv("RX_GPU_HYBRID");
	return d.empty() ? -1 : d[0];
}

// ftIds: the full-text condition's flat id set (ascending row ids), ftRanks: RanksHolder::GetRanksSpan() — MergeInfo::normalizedProc as
// RankT, i.e. integers 0..255 (merger.h:111-140).  outIds / outRanks: the fused list in Merged<desc> order, what mergeRanked builds from
// its hash set.  Returns false — nothing written — when the shape is outside the device kernel's (more than 1024 KNN hits, a rank that is
// not a uint8): the caller then runs the host merge.
template <typename FtIds, typename OutIds, typename OutRanks>
bool FuseRankedOnGpu(int device, const reindexer::Reranker& reranker, bool isUnion, const reindexer::KnnRawResult& knn, const FtIds& ftIds,
					 std::span<const reindexer::RankT> ftRanks, OutIds& outIds, OutRanks& outRanks) {
	using reindexer::IdType;
	std::vector<int32_t> knnIds;
	std::vector<float> knnRanks;
	const bool ok = std::visit(
		[&](const auto& r) {
			using R = std::decay_t<decltype(r)>;
			if constexpr (std::is_same_v<R, reindexer::EmptyKnnRawResult>) {
				return true;
			} else {
				const auto& ids = r.Ids();
				const auto& dists = r.Dists();
				for (size_t i = 0; i < ids.size(); ++i) {
					if constexpr (std::is_same_v<R, reindexer::IvfKnnRawResult>) {
						if (ids[i] < 0) break;   // FAISS pads with -1 (selectiteratorcontainer.cc:1330)
						knnIds.push_back(reindexer::FloatVectorId::FromNumber(ids[i]).RowId().ToNumber());
						knnRanks.push_back(dists[i]);
					} else {
						knnIds.push_back(ids[i].ToNumber());
						knnRanks.push_back(dists[i].Value());
					}
				}
				return knnIds.size() <= 1024;
			}
		},
		knn.AsVariant());
	if (!ok || ftIds.size() != ftRanks.size()) return false;
	std::vector<int32_t> fIds(ftIds.size());
	std::vector<uint8_t> fRanks(ftIds.size());
	for (size_t i = 0; i < ftIds.size(); ++i) {
		const float v = ftRanks[i].Value();
		if (!(v >= 0.f && v <= 255.f) || v != std::floor(v)) return false;
		fIds[i] = ftIds[i].ToNumber();
		fRanks[i] = uint8_t(v);
	}
	rxgpu_hybrid_params hp{};
	hp.is_union = isUnion ? 1 : 0;
	hp.desc = *reranker.Desc() ? 1 : 0;
	if (reranker.IsRRF()) {
		hp.kind = 0;
		hp.params[0] = std::get<reindexer::RerankerRRF>(reranker.AsVariant()).RankConst();   // accessor added by patch 0005 (reranker.h)
	} else {
		hp.kind = 1;
		const auto p = std::get<reindexer::RerankerLinear>(reranker.AsVariant()).Params();
		for (int i = 0; i < 5; ++i) hp.params[i] = p[i];
	}
	const size_t cap = knnIds.size() + fIds.size() + 1;
	std::vector<int32_t> oIds(cap);
	std::vector<float> oRanks(cap);
	uint64_t n = 0;
	if (rxgpu_hybrid_fuse(device, &hp, int(knn.Metric()), knnIds.data(), knnRanks.data(), uint32_t(knnIds.size()), fIds.data(), fRanks.data(),
						  uint32_t(fIds.size()), oIds.data(), oRanks.data(), cap, &n) != RXGPU_OK) {
		return false;
	}
	outIds.reserve(n);
	outRanks.reserve(n);
	for (uint64_t i = 0; i < n; ++i) {
		outIds.push_back(IdType::FromNumber(oIds[i]));
		outRanks.push_back(reindexer::RankT{oRanks[i]});
	}
	return true;
}

}  // namespace rxgpu::host
