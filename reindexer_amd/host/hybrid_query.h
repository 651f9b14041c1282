// One hybrid query — `WHERE ft = '...' OR/AND KNN(vec, [...], k) ORDER BY RRF() / rank expression` (BASELINE configs[4], hybrid.md) — with
// NOTHING leaving HBM between the engines:
//   KNN half   GpuBruteforceMap::SearchKnnResident    the scan's exact top-(k + 1) (dist, row) list stays in the index's device buffers
//   FT half    GpuFtMerger::MergeQueryResident        ft_finish's (document, proc) list stays in the merger's device buffer
//   fusion     GpuFtMerger::PrepareResident           the FT-only part (postProcessResults, order of the documents among themselves, rank
//                                                     classes) right behind the merge — overlapped with the scan
//              GpuFtMerger::FuseResident              the join (MergerRankedImpl / mergeRanked) once both halves are there (hybrid_fuse.hip);
//                                                     the two halves run on their own streams and meet through an event
// One list of (row id, fused rank) in Merged<desc> order comes back.  Mirrors what the planner does with the two SelectKeyResults in
// SelectIteratorContainer (cpp_src/core/nsselecter/selectiteratorcontainer.cc:1305-1559); hybrid_rerank.h is the same fusion on the host.
// A distance tie straddling the k-th place is decided by labels (bruteforce.cc:103-127's strict admission, replayed by
// GpuBruteforceMap::SearchKnn): the device reports it and that query is redone through the host-side pieces — same result, rare.
#pragma once

#include <algorithm>
#include <vector>

#include "gpu_bruteforce_map.h"
#include "gpu_ft_merger.h"
#include "hybrid_rerank.h"
#include "knn_select.h"

namespace rxgpu::host {

// dRowOfDoc: device int32 [totalDocs], vdoc -> row id, when the two do not coincide (IndexText's vdoc table, 1:1 texts); else null
inline HybridFused HybridQueryResident(const GpuBruteforceMap& map, const GpuFtMerger& ft, const FtConfig& cfg, const std::vector<QueryTerm>& terms,
									   const uint8_t* docsExcluded, const float* key, size_t k, const HybridFuseParams& hp, const void* dRowOfDoc = nullptr,
									   const int32_t* hostRowOfDoc = nullptr, const QuerySynonyms* synonyms = nullptr) {
	// HnswIndexBase::search normalises the key for cosine (hnsw_index.cc:166-173)
	std::vector<float> normalized;
	const float* q = key;
	if (map.Metric() == VectorMetric::Cosine) {
		normalized.resize(map.Dim());
		NormalizeCopyVector(key, int32_t(map.Dim()), normalized.data());
		q = normalized.data();
	}
	// The resident KNN list holds at most 128 entries (k + 1 <= 128) and lives on ONE device, the resident merge likewise: a wider k, a sharded
	// mirror or a merger over a device list takes the host-side pieces below straight away (same result; the fusion kernel itself takes k <= 1024)
	const bool residentFits = k + 1 <= 128 && !map.Sharded() && !ft.Sharded();
	HybridFused fused;
	if (residentFits) {
	// FT half first: the merge train and the FT-only part of the fusion (postProcessResults, the sort by id, the class tables) are on the
	// merger's stream before the scan's persistent workgroups fill the chip; they run while the scan — ten times longer — streams the corpus
	if (synonyms && !synonyms->Empty()) {   // multi-word synonyms: the documents Merge() removes stay marked in HBM, the fusion skips them
		ft.MergeQueryResident(cfg, terms, *synonyms, docsExcluded);
	} else {
		ft.MergeQueryResident(cfg, terms, docsExcluded);
	}
	ft.PrepareResident(cfg, hp, int(map.Metric()), dRowOfDoc);
	const GpuBruteforceMap::ResidentKnn knn = map.SearchKnnResident(q, k);            // enqueued on the index's stream
	fused = ft.FuseResident(cfg, hp, int(map.Metric()), knn.dDist, knn.dRow, knn.dCount, knn.entries, uint32_t(std::min<size_t>(k, knn.entries)),
							knn.stream, dRowOfDoc, knn.dRowIds);
	if (!fused.knnBoundaryTie) return fused;
	}
	// the k-th place is a distance tie: the Map's label-aware replay decides it; assemble this query from the host-side pieces
	KnnSearchParams params;
	params.k = k;
#if defined(RXGPU_IN_TREE)
	const ConstFloatVectorView keyView{std::span<const float>{key, map.Dim()}};
#else
	const ConstFloatVectorView keyView{key, map.Dim()};
#endif
	const KnnSelectResult sel = KnnSelectRaw(map, keyView, params, /*isArray*/ false);
	const MergeData md = synonyms && !synonyms->Empty() ? ft.MergeQuery(cfg, terms, *synonyms, docsExcluded, RankSortType::RankAndID)
													   : ft.MergeQuery(cfg, terms, docsExcluded, RankSortType::RankAndID);
	std::vector<int32_t> ftIds(md.size());
	std::vector<float> ftRanks(md.size());
	for (size_t i = 0; i < md.size(); ++i) {
		ftIds[i] = hostRowOfDoc ? hostRowOfDoc[md[i].id] : md[i].id;
		ftRanks[i] = md[i].proc;
	}
	const FtById byId = PrepareFtById(ftIds, ftRanks);
	const HybridMergeType type = hp.isUnion ? HybridMergeType::Union : HybridMergeType::Intersection;
	const HybridResult res = hp.linear ? MergeRankedLinear(RerankerLinear{hp.params[0], hp.params[1], hp.params[2], hp.params[3], hp.params[4]}, type, hp.desc,
														  sel.ids, sel.ranks, byId.ids, byId.ranks)
									   : MergeRankedRRF(RerankerRRF{hp.params[0]}, type, hp.desc, map.Metric(), sel.ids, sel.ranks, byId.ids, byId.positions);
	HybridFused out;
	out.ids = res.ids;
	out.ranks = res.ranks;
	out.knnBoundaryTie = residentFits;   // (set when the resident path handed the query over because of a tie at the k-th place)
	return out;
}

}  // namespace rxgpu::host
