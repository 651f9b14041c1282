// GpuFtMerger — GPU stand-in for ft::Merger<IdCont, MergeData, uint32_t>::Merge<Bm25Rx> on Simple() queries
// (cpp_src/core/ft/ft_fast/merger.h:36-57, mergerimpl.h:466-484 -> mergeSimple :194-250), i.e. the call site
// Selector<IdCont>::mergeResults (selecterimpl.h:611-628) for a query with ONE OR-term and any number of sub-terms
// (original word, typos, translit, stems, ...).  Same inputs (sub-term posting lists + procs, FtDslOpts, FTConfig, docsExcluded,
// DocsStatsGetter), same output (MergeData: vdoc id, proc, field, normalizedProc).
//
// Device: per-posting BM25 ranks, max per document, mergeLimit admission (bm25.hip via rxgpu_ft_merge_simple_raw).
// Host:   the O(mergeLimit) tail of the merger — addFullMatchBoost (merger.h:100-109) and postProcessResults (:111-155).
// Multi-term queries (AND / OR / NOT terms: restricting bitmask, preselect, mergeTerm with position distances,
// mergerimpl.h:107-192, 252-464) go through MergeQuery (ft_merge.hip via rxgpu_ft_merge_terms_raw).
// Phrases run through PhraseMerger::Merge on the device first (ft_phrase.hip via rxgpu_ft_merge_query_raw) and join the merge as query parts
// (mergePhrase, mergerimpl.h:39-90); multi-word synonyms ride behind the parts (rxgpu_ft_merge_query2_raw).  MergeDataAreas (highlight /
// snippet) stays on the reference's CPU merger; Supports() tells the caller which way to go.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <atomic>
#include <vector>

struct rxgpu_ft_index;

namespace rxgpu::host {

// ftconfig.h:118-124
struct FtFieldConfig {
	double bm25Boost = 1.0, bm25Weight = 0.1;
	double termLenBoost = 1.0, termLenWeight = 0.3;
	double positionBoost = 1.0, positionWeight = 0.1;
};
// the FTConfig members the merge reads (ftconfig.h:151-220)
struct FtConfig {
	explicit FtConfig(size_t fieldsCount) : fieldsCfg(fieldsCount) {}
	uint32_t mergeLimit = 20000;
	double distanceBoost = 1.0, distanceWeight = 0.5;
	double bm25k1 = 2.0, bm25b = 0.75;
	// FTConfig::Bm25Config::Bm25Type (ftconfig.h:199-206): the calculator behind Bm25Calculator<BM> (bm25.h:8-68, dispatch selecterimpl.h:615-624);
	// all three are evaluated on the device (ft_rank.hip.h).
	enum class Bm25Type { Classic, Rx, WordCount };
	Bm25Type bm25Type = Bm25Type::Rx;
	double summationRanksByFieldsRatio = 0.0;
	double fullMatchBoost = 1.1;
	int minRank = 5;
	std::vector<FtFieldConfig> fieldsCfg;
};
// ftdsl.h:13-35
struct FtDslFieldOpts {
	float boost = 1.0f;
	bool needSumRank = false;
};
struct FtDslOpts {
	float boost = 1.0f;
	float termLenBoost = 1.0f;
	std::vector<FtDslFieldOpts> fieldsOpts;
};
// querymergedata.h:14-42: one dictionary word matched by the term + its relevancy
struct SubtermRef {
	uint32_t wordId;
	float proc;
	bool suppressed = false;   // SubtermResults::Suppressed() (set by QueryMergeData::SupressDuplicatesInSynonyms, querymergedata.h:221-241)
};
// Synonym<IdCont> (querymergedata.h:178-192): a multi-word synonym = the terms every one of which a document has to hold
struct QueryTerm;
struct QuerySynonyms;
enum class RankSortType { RankOnly, RankAndID, IDOnly, IDAndPositions };   // core/ft/ft_fast/...: how the caller consumes the result
// phrasemerger.h:57-62
struct MergeInfo {
	int32_t id = 0;
	float proc = 0;
	uint8_t field = 0;
	uint8_t normalizedProc = 0;
};
using MergeData = std::vector<MergeInfo>;

// MergeDataAreas<Area> (phrasemerger.h:64-89; core/ft/areaholder.h:9-37, 56-155): what highlight() / snippet() read.  Per merged document an
// entry of vectorAreas (MergeInfoAreas::areaIndex), per field the areas in the order AreasInField::data_ holds them BEFORE Commit().
struct Area {
	uint32_t start = 0, end = 0, arrayIdx = 0;
};
struct FieldAreas {
	std::vector<Area> data;
};
struct MergeInfoAreas {
	int32_t id = 0;
	float proc = 0;
	uint32_t areaIndex = 0xFFFFFFFFu;
	uint8_t field = 0;
	uint8_t normalizedProc = 0;
};
struct MergeDataAreas : std::vector<MergeInfoAreas> {
	std::vector<std::vector<FieldAreas>> vectorAreas;   // [areaIndex][field]
};

// One flattened posting list (IdRelVec of one dictionary word)
struct FlatPostings {
	std::vector<uint32_t> doc, entOff{0};
	std::vector<uint8_t> entField;
	std::vector<uint32_t> entTf, entFirstPos;
	// append one IdRelType: positions given as (field, pos) pairs sorted by field then pos (idrelset.h:14-32 ordering)
	void Add(uint32_t vdoc, const std::pair<uint32_t, uint32_t>* fieldPos, size_t count);
};

// One posting list WITH positions (what mergeTerm needs): PosType words pos | arrayIdx << 28 | field << 56 (idrelset.h:14-32)
struct PositionPostings {
	std::vector<uint32_t> doc, posOff{0};
	std::vector<uint64_t> fpos;
	static uint64_t Pos(uint32_t pos, uint32_t field, uint32_t arrayIdx = 0) noexcept {
		return uint64_t(pos) | (uint64_t(arrayIdx) << 28) | (uint64_t(field) << 56);
	}
	// append one IdRelType: positions sorted ascending (IdRelType::SortAndUnique)
	void Add(uint32_t vdoc, const uint64_t* positions, size_t count) {
		doc.push_back(vdoc);
		fpos.insert(fpos.end(), positions, positions + count);
		posOff.push_back(uint32_t(fpos.size()));
	}
	// Decode a PackedIdRelVec byte stream (cpp_src/core/ft/idrelset.h:155-280: base-128 varints, ids and fields delta-coded against the
	// previous element; elements stored before byte `arrayFoundPos` use the format without array indexes — idrelset.cc:8-235) and
	// append its postings.  Throws std::invalid_argument on a truncated / malformed stream.
	void AppendPacked(const uint8_t* data, size_t len, size_t arrayFoundPos);
};
enum class OpType { Or = 1, And = 2, Not = 3 };   // core/type_consts.h
// TermResults (querymergedata.h:46-98): a query term, its FtDslOpts and the dictionary words it matched
struct QueryTerm {
	OpType op = OpType::Or;
	FtDslOpts opts;
	std::vector<SubtermRef> subterms;
	// FtDslOpts::phraseNum / distance (ftdsl.h:13-35): consecutive terms with the same phraseNum >= 0 are one phrase (PhraseResults,
	// querymergedata.h:100-142; built in selecterimpl.h:482-572), merged by PhraseMerger on the device (ft_phrase.hip)
	int phraseNum = -1;
	int distance = 1;
};

// QueryMergeData::synonyms + PhraseOrTerm::SynonymsIds (querymergedata.h:145-176, 191-193)
struct QuerySynonyms {
	std::vector<std::vector<QueryTerm>> synonyms;      // synonym -> its terms (Synonym::Terms())
	std::vector<std::vector<uint32_t>> partSynonyms;   // query part (a term or a whole phrase) -> ids of its synonyms; may be shorter than the parts
	bool Empty() const noexcept { return synonyms.empty(); }
};

// The hybrid rank fusion on the device (hybrid_fuse.hip): reranker + join type, as MergerRankedImpl gets them (selectiteratorcontainer.cc:1305-1341)
struct HybridFuseParams {
	bool linear = false;   // false: RRF, params[0] = rank_const (60); true: RerankerLinear, params = kKnn, knnDefault, kFt, ftDefault, c
	bool isUnion = true;   // OR between the two ranked conditions (AND: intersection)
	bool desc = true;
	double params[5] = {60.0, 0.0, 0.0, 0.0, 0.0};
};
struct HybridFused {
	std::vector<int32_t> ids;    // row ids in Merged<desc> order
	std::vector<float> ranks;
	bool knnBoundaryTie = false;   // the k-th and (k+1)-th KNN distances are equal: the Map's label-aware replay has to decide the k-th place
};

class GpuFtMerger {
public:
	GpuFtMerger(size_t numFields, int device = 0);
	// SURVEY 8(e) "BM25": the same merger over a DEVICE LIST — the index is cut into document-range shards (rxgpu_ft_create_sharded: every
	// device holds the posting fragments of its documents, idf from the global N / df; the pre-score histograms and the admission table meet
	// in one all-gather each) and every merge returns the single-device result bit for bit.  Queries of terms and multi-word synonyms
	// (what a synonym decides — its mask, the term count, "only parts of it" — concerns one document, and a document lies in one shard)
	// and of phrases (every shard runs PhraseMerger over its fragments; the admission cut, phrasemerger.h:341, is settled between the
	// shards), MergeQueryAreas included; resident (hybrid) merges need a single-device merger; MergeQueryBatch runs its merges one after
	// the other there (each the single index's result).
	GpuFtMerger(size_t numFields, std::vector<int> devices);
	~GpuFtMerger();
	bool Sharded() const noexcept { return sharded_; }
	// ranges of the fullest shard / ranges of an even cut: an index that grew through step commits keeps its cut, the new document ranges
	// pile up on the last shard (rxgpu_ft_shard_imbalance).  1.0 for an unsharded merger.
	double ShardImbalance() const noexcept;
	rxgpu_ft_index* DeviceIndex() const noexcept { return dev_; }
	// (every query shape the single-device merger takes is merged over a device list as well — phrases: PhraseMerger per shard, the admission
	// cut of the whole index settled between the shards; areas: built where the document lies, at its global merge slot — except the
	// resident, hybrid, form: hybrid_query.h fuses on the host then)
	bool ShardedSupports(bool /*hasPhrases*/, bool /*hasSynonyms*/, int /*maxAreasInDoc*/ = 0) const noexcept { return true; }
	GpuFtMerger(const GpuFtMerger&) = delete;

	// IndexText side (CommitFulltext): vdoc statistics and posting lists
	void SetDocs(size_t totalDocs, const float* wordsInField, const float* avgWords, const uint8_t* removed);
	void SetWord(uint32_t wordId, const FlatPostings& postings);

	void SetWord(uint32_t wordId, const PositionPostings& postings);   // usable by Merge and MergeQuery
	// IndexText side, bulk: the dictionary's posting lists as the reference stores them (PackedIdRelVec byte streams, idrelset.h:155-280)
	// are uploaded in one piece and decoded ON THE DEVICE (ft_packed.hip, one thread per word) into everything Merge / MergeQuery read.
	// A single stream is serial work (no sync points in the format), so streams of `hostDecodeFromBytes` bytes or more are decoded by
	// AppendPacked on the host and uploaded as before; the rest — the bulk of a dictionary — never touch the host decoder.
	struct PackedWord {
		uint32_t wordId;
		const uint8_t* data;
		size_t len, arrayFoundPos;   // PackedIdRelVec::arrayFoundPos_ (>= len: no element carries array indexes)
	};
	void SetWordsPacked(const std::vector<PackedWord>& words, size_t hostDecodeFromBytes = size_t(1) << 18);
	// reads a word's device arrays back (tests, diagnostics)
	void GetWord(uint32_t wordId, PositionPostings& positions, FlatPostings& entries, std::vector<uint32_t>& rangeOff) const;

	static bool Supports(size_t numQueryParts, bool hasPhrases, bool hasSynonyms) noexcept {
		(void)hasPhrases;    // PhraseMerger runs on the device
		(void)hasSynonyms;   // multi-word synonyms too (MergeQuery with QuerySynonyms)
		return numQueryParts >= 1;
	}
	static bool Supports(const FtConfig& cfg, size_t numQueryParts, bool hasPhrases, bool hasSynonyms) noexcept {
		(void)cfg;   // every Bm25Type is evaluated on the device
		return Supports(numQueryParts, hasPhrases, hasSynonyms);
	}

	// Merger::Merge<Bm25T> for a Simple() query (Bm25T from cfg.bm25Type)
	MergeData Merge(const FtConfig& cfg, const FtDslOpts& termOpts, std::vector<SubtermRef> subterms, const uint8_t* docsExcluded,
					RankSortType rankSortType) const;

	// Merger::Merge<Bm25T> for any query made of terms and phrases (no multi-word synonyms); one OR/AND term -> Merge()
	MergeData MergeQuery(const FtConfig& cfg, std::vector<QueryTerm> terms, const uint8_t* docsExcluded, RankSortType rankSortType,
						 bool* preselected = nullptr) const;
	// ... with multi-word synonyms (mergerimpl.h:347-361, 393-397, 509-555)
	MergeData MergeQuery(const FtConfig& cfg, std::vector<QueryTerm> terms, QuerySynonyms synonyms, const uint8_t* docsExcluded, RankSortType rankSortType,
						 bool* preselected = nullptr) const;

	// Merger<IdCont, MergeDataAreas<Area>, ..>::Merge (merger.h:36-57, addAreas :196-204): the merge behind highlight() / snippet() on the device
	// (rxgpu_ft_merge_query_areas_raw) for queries of plain terms, Simple() included; maxAreasInDoc = FTConfig::maxAreasInDoc >= 1.
	// Phrases (their areas come out of the PhraseMerger's position chains), multi-word synonyms and AreaDebug: the CPU merger (SupportsAreas).
	static bool SupportsAreas(size_t numQueryParts, bool hasPhrases, bool hasSynonyms, int maxAreasInDoc) noexcept {
		return numQueryParts >= 1 && !hasPhrases && !hasSynonyms && maxAreasInDoc >= 1 && maxAreasInDoc <= 4096;
	}
	MergeDataAreas MergeQueryAreas(const FtConfig& cfg, std::vector<QueryTerm> terms, const uint8_t* docsExcluded, RankSortType rankSortType, int maxAreasInDoc,
								   bool* preselected = nullptr) const;

	// Several queries in hand (the hybrid path's batch, a combiner in front of T planner threads): ONE launch train on the device for all of
	// them (rxgpu_ft_merge_batch_raw: the merge kernels run with the query as the second grid dimension) — the result of query i is what
	// MergeQuery(cfg, queries[i], ...) returns, bit for bit.  docsExcluded: per query, or empty (none); preselected: per query, may be null.
	std::vector<MergeData> MergeQueryBatch(const FtConfig& cfg, std::vector<std::vector<QueryTerm>> queries, const std::vector<const uint8_t*>& docsExcluded,
										   RankSortType rankSortType, std::vector<uint8_t>* preselected = nullptr) const;

	// Hybrid query, FT half: the same merge, but the result STAYS IN HBM (no export, no wait).  False when the query merges nothing
	// (Empty(), no sub-terms) — there is then no resident result and FuseResident sees an empty FT side.
	bool MergeQueryResident(const FtConfig& cfg, std::vector<QueryTerm> terms, const uint8_t* docsExcluded) const;
	// ... with multi-word synonyms (round 4): the documents Merge() removes with their partial synonym stay marked in HBM, the fusion skips them
	bool MergeQueryResident(const FtConfig& cfg, std::vector<QueryTerm> terms, QuerySynonyms synonyms, const uint8_t* docsExcluded) const;
	// ... the FT-only half of the fusion (postProcessResults, the documents' mutual order, rank-class tables) enqueued behind that merge:
	// called BEFORE the KNN search is started it runs while the scan streams the corpus (optional: FuseResident does it when it was not) ...
	void PrepareResident(const FtConfig& cfg, const HybridFuseParams& hp, int metric, const void* dRowOfDoc = nullptr) const;
	// ... and the fusion: postProcessResults + MergerRankedImpl on the device over the resident merge and a KNN result that lies in HBM
	// as rxgpu_search_knn_device left it ((dist, row) best first; the first k take part; knnStream = the stream of that search).
	HybridFused FuseResident(const FtConfig& cfg, const HybridFuseParams& hp, int metric, const void* dKnnDist, const void* dKnnRow, const void* dKnnCount,
							 uint32_t knnEntries, uint32_t k, void* knnStream, const void* dRowOfDoc = nullptr, const void* dRowIdOfRow = nullptr) const;

	size_t TotalDocs() const noexcept { return totalDocs_; }
	void ReadStats(uint64_t& postings, double& kernelMs) const;
	// SetWordsPacked: device time of the two decode kernels, stream bytes read per pass, array bytes produced, since the last call
	void ReadPackedStats(double& countMs, double& writeMs, uint64_t& bytesIn, uint64_t& bytesOut) const;
	// ... and the wall time spent inside the library's packed-upload calls (gather, upload, both passes, the dictionary entries), since the last call
	void ReadPackedWall(double& wallMs) const;
	// FuseResident: fusions, the device time of their join kernel (critical path) and of the overlapped prepare kernel since the last call
	void ReadFuseStats(uint64_t& calls, double& kernelMs, double* prepareMs = nullptr) const;
	// wall time spent inside Merge / MergeQuery since the last call (everything behind the Merger boundary: plan, launches, the wait,
	// unpacking, postProcessResults) and the number of calls; resets both
	void ReadTiming(uint64_t& calls, double& totalMs) const;

private:
	MergeData mergeImpl(const FtConfig& cfg, const FtDslOpts& termOpts, std::vector<SubtermRef> subterms, const uint8_t* docsExcluded,
						RankSortType rankSortType, bool resident) const;
	MergeData mergeQueryImpl(const FtConfig& cfg, std::vector<QueryTerm> terms, const uint8_t* docsExcluded, RankSortType rankSortType, bool* preselected,
							 bool resident, QuerySynonyms* synonyms = nullptr) const;
	void postProcess(const FtConfig& cfg, MergeData& out, RankSortType rankSortType) const;
	void postProcess(const FtConfig& cfg, MergeDataAreas& out, RankSortType rankSortType) const;
	const size_t numFields_;
	size_t totalDocs_ = 0;
	std::vector<float> words_;   // host copy for addFullMatchBoost
	rxgpu_ft_index* dev_ = nullptr;
	bool sharded_ = false;
	mutable std::atomic<uint64_t> timedCalls_{0}, timedNs_{0};
};

}  // namespace rxgpu::host
