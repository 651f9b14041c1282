// In-tree adapter (RXGPU_IN_TREE only) between the reference's ft_fast selector and GpuFtMerger: the FT half of the drop-in boundary.
//
//   Selector<IdCont>::mergeResults (cpp_src/core/ft/ft_fast/selecterimpl.h:608-628) constructs ft::Merger and calls Merge<Bm25T>;
//   integration/patches/0003-ft-fast-gpu-merger.patch puts `rxgpu::host::TryMergeOnGpu(...)` in front of it.  Everything that call needs is
//   converted here from the reference's OWN types:
//     FTConfig (core/ft/config/ftconfig.h:118-220)                         -> FtConfig            (ToGpuCfg)
//     FtDslOpts (core/ft/ftdsl.h:13-35)                                    -> FtDslOpts           (ToGpuOpts)
//     ft::QueryMergeData<IdCont> (core/ft/ft_fast/querymergedata.h:14-242) -> std::vector<QueryTerm>  (ToGpuTerms; word id = WordIdType::b.id,
//                                                                           the index of DataHolder<IdCont>::words_, dataholder.h:186-207)
//     FtMergeStatuses::Statuses (core/index/ft_preselect.h:10-17)          -> one byte per vdoc, or nothing when no bit is set
//     MergeData (ours) -> ft::MergeData (phrasemerger.h:57-78)             (ToRxMergeData)
//   and the posting lists are handed over at commit time (IndexText::commitFulltextImpl, core/index/indextext/indextext.cc:817-885):
//     DataHolder<PackedIdRelVec>::words_ — the raw varint streams, decoded ON THE DEVICE (GpuFtMerger::SetWordsPacked); needs the three
//                                          accessors the patch adds to PackedIdRelVec (RawData / RawSize / ArrayFoundPos)
//     DataHolder<IdRelVec>::words_       — IdRelType by IdRelType through PositionPostings::Add
//     vdoc statistics                    — the duck-typed DocsStatsGetter (indextext.h:245-258): DocRemoved / NumWordsInField / AvgWordsCount
//   Only the words whose list changed since the last commit travel again (fingerprint: byte size + FNV-1a of the stream / (size, last id)).
//
// What still goes to the reference's CPU merger (TryMergeOnGpu returns false): MergeDataAreas (highlight / snippet) — the patched
// mergeResults only branches for plain ft::MergeData.  Phrases (PhraseMerger as kernels, ft_phrase.hip) and multi-word synonyms go to the device.
#pragma once
#include <cstdio>
#if !defined(RXGPU_IN_TREE)
#error "rx_ft_seam.h is for the build inside cpp_src (define RXGPU_IN_TREE)"
#endif

#include <cstdlib>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "core/enums.h"
#include "core/ft/config/ftconfig.h"
#include "core/ft/ft_fast/dataholder.h"
#include "core/ft/ft_fast/phrasemerger.h"
#include "core/ft/ft_fast/querymergedata.h"
#include "core/ft/ftdsl.h"
#include "core/ft/idrelset.h"
#include "core/index/ft_preselect.h"
#include "device_list.h"
#include "gpu_ft_merger.h"

namespace rxgpu::host {

// RX_GPU_FT_INDEXES=<device list> (device_list.h: "3", "0,1,2,3", "0-7") routes the merge step of `text` (ft_fast) indexes to the MI355X
// engine (unset / empty / malformed: the CPU merger).  More than one device: the index is cut into document-range shards (SURVEY 8e "BM25").
inline std::vector<int> GpuFtDevicesFromEnv() { return GpuDevicesFromEnv("RX_GPU_FT_INDEXES"); }
inline int GpuFtDeviceFromEnv() noexcept {
	const std::vector<int> d = GpuFtDevicesFromEnv();
	return d.empty() ? -1 : d[0];
}

inline FtConfig ToGpuCfg(const reindexer::FTConfig& c) {
	FtConfig g(c.fieldsCfg.size());
	g.mergeLimit = c.mergeLimit;
	g.distanceBoost = c.distanceBoost;
	g.distanceWeight = c.distanceWeight;
	g.bm25k1 = c.bm25Config.bm25k1;
	g.bm25b = c.bm25Config.bm25b;
	using RT = reindexer::FTConfig::Bm25Config::Bm25Type;
	g.bm25Type = c.bm25Config.bm25Type == RT::rx ? FtConfig::Bm25Type::Rx
											   : (c.bm25Config.bm25Type == RT::classic ? FtConfig::Bm25Type::Classic : FtConfig::Bm25Type::WordCount);
	g.summationRanksByFieldsRatio = c.summationRanksByFieldsRatio;
	g.fullMatchBoost = c.fullMatchBoost;
	g.minRank = c.minRank;
	for (size_t f = 0; f < c.fieldsCfg.size(); ++f) {
		const reindexer::FTFieldConfig& s = c.fieldsCfg[f];
		g.fieldsCfg[f] = FtFieldConfig{s.bm25Boost, s.bm25Weight, s.termLenBoost, s.termLenWeight, s.positionBoost, s.positionWeight};
	}
	return g;
}

inline FtDslOpts ToGpuOpts(const reindexer::FtDslOpts& o) {
	FtDslOpts g;
	g.boost = o.boost;
	g.termLenBoost = o.termLenBoost;
	g.fieldsOpts.resize(o.fieldsOpts.size());
	for (size_t f = 0; f < o.fieldsOpts.size(); ++f) g.fieldsOpts[f] = FtDslFieldOpts{o.fieldsOpts[f].boost, o.fieldsOpts[f].needSumRank};
	return g;
}

inline bool ToGpuSortType(reindexer::RankSortType t, RankSortType& out) noexcept {
	switch (t) {
		case reindexer::RankSortType::RankOnly: out = RankSortType::RankOnly; return true;
		case reindexer::RankSortType::RankAndID: out = RankSortType::RankAndID; return true;
		case reindexer::RankSortType::IDOnly: out = RankSortType::IDOnly; return true;
		case reindexer::RankSortType::IDAndPositions: out = RankSortType::IDAndPositions; return true;
		case reindexer::RankSortType::ExternalExpression: return false;   // merger.h:151: the CPU merger throws errLogic — let it
	}
	return false;
}

// The query as the merger walks it: queryParts in order, sub-terms in SortSubterms() order (the caller has sorted them, mergerimpl.h:479).
// A phrase part (PhraseResults, querymergedata.h:100-142) travels as its terms, marked with one phrase number and each with its
// FtDslOpts::distance — what GpuFtMerger::MergeQuery groups again (PhraseMerger on the device).
// False when the query holds something the GPU merger does not evaluate (multi-word synonyms).
template <typename IdCont>
bool ToGpuTerm(const reindexer::ft::TermResults<IdCont>& t, int phraseNum, std::vector<QueryTerm>& terms) {
	QueryTerm g;
	switch (t.Op()) {
		case OpOr: g.op = OpType::Or; break;
		case OpAnd: g.op = OpType::And; break;
		case OpNot: g.op = OpType::Not; break;
		default: return false;
	}
	g.opts = ToGpuOpts(t.Opts());
	g.phraseNum = phraseNum;
	g.distance = t.Distance();
	g.subterms.reserve(t.NumSubterms());
	for (const auto& st : t) g.subterms.push_back(SubtermRef{uint32_t(st.PatternID().b.id), st.Proc(), st.Suppressed()});
	terms.push_back(std::move(g));
	return true;
}
// Multi-word synonyms (QueryMergeData::synonyms, PhraseOrTerm::SynonymsIds) -> QuerySynonyms; the Suppressed() marks of
// SupressDuplicatesInSynonyms (the selecter has called it, selecterimpl.h:606) travel with the sub-terms.
template <typename IdCont>
bool ToGpuTerms(reindexer::ft::QueryMergeData<IdCont>& q, std::vector<QueryTerm>& terms, bool* hasPhrases = nullptr, QuerySynonyms* synonyms = nullptr) {
	if (hasPhrases) *hasPhrases = false;
	if (!q.synonyms.empty() && !synonyms) return false;
	if (synonyms) {
		synonyms->synonyms.clear();
		synonyms->partSynonyms.clear();
		for (auto& syn : q.synonyms) {
			synonyms->synonyms.emplace_back();
			for (const auto& t : syn.Terms()) {
				if (!ToGpuTerm(t, -1, synonyms->synonyms.back())) return false;
			}
		}
	}
	terms.clear();
	terms.reserve(q.queryParts.size());
	int phraseNum = 0;
	for (auto& qp : q.queryParts) {
		if (!qp.SynonymsIds().empty() && !synonyms) return false;
		if (synonyms) {
			synonyms->partSynonyms.emplace_back();
			for (size_t id : qp.SynonymsIds()) synonyms->partSynonyms.back().push_back(uint32_t(id));
		}
		if (qp.IsTerm()) {
			if (!ToGpuTerm(qp.Term(), -1, terms)) return false;
			continue;
		}
		auto& ph = qp.Phrase();   // (PhraseResults::Term has no const overload)
		if (ph.NumTerms() < 2) return false;   // FtDSLQuery::closeGroup (ftdsl.cc:87-101) marks groups of two or more terms only
		for (size_t i = 0; i < ph.NumTerms(); ++i) {
			if (!ToGpuTerm(ph.Term(i), phraseNum, terms)) return false;
		}
		++phraseNum;
		if (hasPhrases) *hasPhrases = true;
	}
	return true;
}

inline void ToRxMergeData(const MergeData& in, reindexer::ft::MergeData& out) {
	out.resize(in.size());
	for (size_t i = 0; i < in.size(); ++i) {
		reindexer::ft::MergeInfo& o = out[i];
		o.id = reindexer::IdType::FromNumber(in[i].id);
		o.proc = in[i].proc;
		o.field = in[i].field;
		o.normalizedProc = in[i].normalizedProc;
	}
}

// The device mirror of one ft_fast index: owned by the DataHolder (the patch adds `std::shared_ptr<GpuFtMirror> gpuMirror_` to IDataHolder),
// refreshed by SyncGpuFtMirror at the end of every commit, read by TryMergeOnGpu under the index's shared lock.
class GpuFtMirror {
public:
	GpuFtMirror(size_t numFields, int device) : merger_(numFields, device), numFields_(numFields) {}
	GpuFtMirror(size_t numFields, std::vector<int> devices) : merger_(numFields, std::move(devices)), numFields_(numFields) {}

	const GpuFtMerger& Merger() const noexcept { return merger_; }
	size_t SyncedWords() const noexcept { return prints_.size(); }
	size_t SyncedDocs() const noexcept { return merger_.TotalDocs(); }
	size_t NumFields() const noexcept { return numFields_; }

	// vdoc statistics (b7) from the reference's DocsStatsGetter; vdoc 0 is the empty sentinel like everywhere in ft_fast
	template <typename DocsStatsGetter>
	void SyncDocs(size_t totalDocs, const DocsStatsGetter& stats) {
		std::vector<float> words(totalDocs * numFields_), avg(numFields_);
		std::vector<uint8_t> removed(totalDocs);
		for (size_t d = 0; d < totalDocs; ++d) {
			removed[d] = stats.DocRemoved(uint32_t(d)) ? 1 : 0;
			// the merger reads size_t NumWordsInField() (indextext.h:249-253: the float count truncated) — hand over the same number
			if (!removed[d]) {
				for (size_t f = 0; f < numFields_; ++f) words[d * numFields_ + f] = float(stats.NumWordsInField(uint32_t(d), uint32_t(f)));
			}
		}
		for (size_t f = 0; f < numFields_; ++f) avg[f] = stats.AvgWordsCount(uint32_t(f));
		merger_.SetDocs(totalDocs, words.data(), avg.data(), removed.data());
	}

	// Optimization::Memory: the packed streams travel as they are and are decoded on the device
	void SyncWords(const std::vector<reindexer::PackedWordEntry<reindexer::PackedIdRelVec>>& words) {
		std::vector<GpuFtMerger::PackedWord> changed;
		prints_.resize(words.size());
		for (size_t w = 0; w < words.size(); ++w) {
			const reindexer::PackedIdRelVec& v = words[w].vids;
			const Print p{v.RawSize(), fnv1a(v.RawData(), v.RawSize())};
			if (p == prints_[w]) continue;
			prints_[w] = p;
			changed.push_back(GpuFtMerger::PackedWord{uint32_t(w), v.RawData(), v.RawSize(), v.ArrayFoundPos()});
		}
		if (!changed.empty()) merger_.SetWordsPacked(changed);
	}

	// Optimization::CPU: plain vectors of IdRelType
	void SyncWords(const std::vector<reindexer::PackedWordEntry<reindexer::IdRelVec>>& words) {
		prints_.resize(words.size());
		std::vector<uint64_t> fpos;
		for (size_t w = 0; w < words.size(); ++w) {
			const reindexer::IdRelVec& v = words[w].vids;
			// The whole list is hashed, as for the packed streams: a re-commit erases the last step's words and rebuilds them at the same
			// indices (dataholder.cc:116), so a DIFFERENT word — or the same word with changed earlier documents, positions or fields — can
			// land on index w with the same length and the same last entry.
			uint64_t hsh = kFnvBasis;
			for (const reindexer::IdRelType& e : v) {
				hsh = mix(hsh, uint64_t(e.Id()) | (uint64_t(e.Pos().size()) << 32));
				for (const reindexer::PosType& pos : e.Pos()) hsh = mix(hsh, PositionPostings::Pos(pos.pos(), pos.field(), pos.arrayIdx()));
			}
			const Print p{v.size(), hsh};
			if (p == prints_[w]) continue;
			prints_[w] = p;
			PositionPostings pp;
			for (const reindexer::IdRelType& e : v) {
				fpos.clear();
				for (const reindexer::PosType& pos : e.Pos()) fpos.push_back(PositionPostings::Pos(pos.pos(), pos.field(), pos.arrayIdx()));
				pp.Add(e.Id(), fpos.data(), fpos.size());
			}
			merger_.SetWord(uint32_t(w), pp);
		}
	}

private:
	struct Print {
		uint64_t size = ~uint64_t(0), hash = 0;
		bool operator==(const Print& o) const noexcept { return size == o.size && hash == o.hash; }
	};
	static constexpr uint64_t kFnvBasis = 1469598103934665603ull;
	static uint64_t mix(uint64_t h, uint64_t x) noexcept {   // FNV-1a over 64-bit words, folded once more so that the high bits take part
		h = (h ^ x) * 1099511628211ull;
		return h ^ (h >> 29);
	}
	static uint64_t fnv1a(const uint8_t* p, size_t n) noexcept {
		uint64_t h = kFnvBasis;
		for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
		return h;
	}
	GpuFtMerger merger_;
	const size_t numFields_;
	std::vector<Print> prints_;
};

// End of IndexText::commitFulltextImpl: (re)creates the mirror when the engine is switched on and brings it up to date.
template <typename IdCont, typename DocsStatsGetter>
void SyncGpuFtMirror(reindexer::DataHolder<IdCont>& holder, std::shared_ptr<GpuFtMirror>& mirror, size_t totalDocs, size_t numFields,
					 const DocsStatsGetter& stats) {
	std::vector<int> devices = GpuFtDevicesFromEnv();
	if (devices.empty()) {
		mirror.reset();
		return;
	}
	// A mirror over a device list keeps its document-range cut while the index grows through step commits: the new ranges pile up on the last
	// shard.  Once that shard holds twice its share the mirror is built again (every word uploaded once more, an even cut).
	const bool recut = mirror && mirror->Merger().ShardImbalance() > 2.0;
	try {
		if (!mirror || recut || holder.status_ == reindexer::FullRebuild) mirror = std::make_shared<GpuFtMirror>(numFields, std::move(devices));
		mirror->SyncDocs(totalDocs, stats);
		mirror->SyncWords(holder.GetWords());   // DataHolder<IdCont>::words_ (dataholder.h:186-207)
	} catch (const std::exception& e) {
		// the commit must not fail because the device copy could not follow: without a mirror TryMergeOnGpu declines and the CPU merger runs
		std::fprintf(stderr, "rxgpu: ft_fast device mirror dropped, the CPU merger takes over: %s\n", e.what());
		mirror.reset();
	}
}

// Selector<IdCont>::mergeResults, GPU branch.  Returns false — and leaves everything untouched — when the CPU merger has to run.
// MergeDataAreas<Area> back in the reference's types: MergeInfoAreas + one AreasInDocument per merged document, every field's areas adopted as
// the merge left them (AreasInField::AdoptRaw, patch 0003: data_ in insertion order, not committed — GetAreas() sorts and joins them on first use)
inline void ToRxMergeDataAreas(MergeDataAreas&& in, size_t fieldSize, reindexer::ft::MergeDataAreas<reindexer::Area>& out) {
	out.resize(in.size());
	for (size_t i = 0; i < in.size(); ++i) {
		reindexer::ft::MergeInfoAreas& o = out[i];
		o.id = reindexer::IdType::FromNumber(in[i].id);
		o.proc = in[i].proc;
		o.field = in[i].field;
		o.normalizedProc = in[i].normalizedProc;
		o.areaIndex = in[i].areaIndex;
	}
	out.vectorAreas.resize(in.vectorAreas.size());
	for (size_t d = 0; d < in.vectorAreas.size(); ++d) {
		reindexer::AreasInDocument<reindexer::Area>& doc = out.vectorAreas[d];
		doc.ReserveField(int(fieldSize));   // addDoc (merger.h:165-166)
		for (size_t f = 0; f < in.vectorAreas[d].size() && f < fieldSize; ++f) {
			const std::vector<Area>& src = in.vectorAreas[d][f].data;
			if (src.empty()) continue;
			reindexer::h_vector<reindexer::Area, 2> data;
			data.reserve(src.size());
			for (const Area& a : src) data.emplace_back(a.start, a.end, a.arrayIdx);
			doc.GetAreasRaw(unsigned(f))->AdoptRaw(std::move(data), int(src.size()));
		}
	}
}

template <typename IdCont, typename MergedDataType>
bool TryMergeOnGpu(const GpuFtMirror* mirror, const reindexer::FTConfig& cfg, size_t totalNumDocs, reindexer::ft::QueryMergeData<IdCont>& q,
				   reindexer::RankSortType rankSortType, const reindexer::FtMergeStatuses::Statuses& docsExcluded, bool inTransaction,
				   MergedDataType& result, int maxAreasInDoc = 0) {
	if constexpr (std::is_same_v<MergedDataType, reindexer::ft::MergeDataAreas<reindexer::Area>>) {
		// highlight() / snippet(): queries of plain terms run on the device (GpuFtMerger::MergeQueryAreas); phrases, multi-word synonyms and an
		// unlimited maxAreasInDoc go to the CPU merger
		(void)inTransaction;
		if (!mirror || mirror->SyncedDocs() != totalNumDocs) return false;
		RankSortType sortType;
		if (!ToGpuSortType(rankSortType, sortType)) return false;
		if (q.Empty()) return false;
		q.SortSubterms();
		std::vector<QueryTerm> terms;
		bool hasPhrases = false;
		QuerySynonyms synonyms;
		if (!ToGpuTerms(q, terms, &hasPhrases, &synonyms)) return false;
		if (!GpuFtMerger::SupportsAreas(terms.size(), hasPhrases, !q.synonyms.empty(), maxAreasInDoc)) return false;
		if (!mirror->Merger().ShardedSupports(hasPhrases, !q.synonyms.empty(), maxAreasInDoc)) return false;
		std::vector<uint8_t> excluded;
		const uint8_t* excludedPtr = nullptr;
		if (docsExcluded.PopCount() != 0) {
			excluded.resize(totalNumDocs);
			for (size_t d = 0; d < totalNumDocs && d < docsExcluded.size(); ++d) excluded[d] = docsExcluded[d] ? 1 : 0;
			excludedPtr = excluded.data();
		}
		MergeDataAreas merged = mirror->Merger().MergeQueryAreas(ToGpuCfg(cfg), std::move(terms), excludedPtr, sortType, maxAreasInDoc);
		ToRxMergeDataAreas(std::move(merged), mirror->NumFields(), result);
		return true;
	} else if constexpr (!std::is_same_v<MergedDataType, reindexer::ft::MergeData>) {
		return false;   // MergeDataAreas<AreaDebug> (debug_rank strings) is built by the CPU merger
	} else {
		(void)maxAreasInDoc;
		(void)inTransaction;   // only gates ThrowOnCancel checkpoints in the CPU merger (mergerimpl.h:118, 200); one GPU merge is a fraction of a millisecond
		if (!mirror || mirror->SyncedDocs() != totalNumDocs) return false;
		RankSortType sortType;
		if (!ToGpuSortType(rankSortType, sortType)) return false;
		if (q.Empty()) return false;   // the CPU merger returns its empty result
		q.SortSubterms();   // Merge() does it before anything reads the sub-terms (mergerimpl.h:479)
		std::vector<QueryTerm> terms;
		bool hasPhrases = false;
		QuerySynonyms synonyms;
		if (!ToGpuTerms(q, terms, &hasPhrases, &synonyms)) return false;
		if (!GpuFtMerger::Supports(terms.size(), hasPhrases, !q.synonyms.empty())) return false;
		if (!mirror->Merger().ShardedSupports(hasPhrases, !q.synonyms.empty())) return false;
		std::vector<uint8_t> excluded;
		const uint8_t* excludedPtr = nullptr;
		if (docsExcluded.PopCount() != 0) {
			excluded.resize(totalNumDocs);
			for (size_t d = 0; d < totalNumDocs && d < docsExcluded.size(); ++d) excluded[d] = docsExcluded[d] ? 1 : 0;
			excludedPtr = excluded.data();
		}
		const MergeData merged = synonyms.Empty() ? mirror->Merger().MergeQuery(ToGpuCfg(cfg), std::move(terms), excludedPtr, sortType)
												  : mirror->Merger().MergeQuery(ToGpuCfg(cfg), std::move(terms), std::move(synonyms), excludedPtr, sortType);
		ToRxMergeData(merged, result);
		return true;
	}
}

}  // namespace rxgpu::host
