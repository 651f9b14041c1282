#include "gpu_ft_merger.h"

#include <chrono>
#include <algorithm>
#include <stdexcept>

#include "rxgpu.h"

namespace rxgpu::host {

namespace {
[[noreturn]] void throwDevice(const char* what) { throw std::runtime_error(std::string(what) + ": " + rxgpu_last_error()); }
}  // namespace

void FlatPostings::Add(uint32_t vdoc, const std::pair<uint32_t, uint32_t>* fieldPos, size_t count) {
	doc.push_back(vdoc);
	size_t i = 0;
	while (i < count) {   // group by field like calcTermRankImpl (phrasemergerimpl.h:24-38)
		const uint32_t f = fieldPos[i].first;
		const size_t begin = i;
		while (i < count && fieldPos[i].first == f) ++i;
		entField.push_back(uint8_t(f));
		entTf.push_back(uint32_t(i - begin));
		entFirstPos.push_back(fieldPos[begin].second);
	}
	entOff.push_back(uint32_t(entField.size()));
}

namespace {
// tools/varint.h:122-176: base-128 varint, at most 5 bytes for a uint32 (the 5th byte carries bits 28..31 unmasked)
uint32_t readVarint(const uint8_t*& p, const uint8_t* end) {
	uint32_t v = 0;
	for (unsigned i = 0; i < 5; ++i) {
		if (p == end) throw std::invalid_argument("PackedIdRelVec: truncated varint");
		const uint8_t b = *p++;
		if (i == 4) return v | (uint32_t(b) << 28);
		v |= uint32_t(b & 0x7f) << (7 * i);
		if (!(b & 0x80)) return v;
	}
	return v;
}
}  // namespace

void PositionPostings::AppendPacked(const uint8_t* data, size_t len, size_t arrayFoundPos) {
	const uint8_t* p = data;
	const uint8_t* const end = data + len;
	uint32_t lastId = 0, lastField = 0;   // PackedIdRelVec::state (idrelset.h:166-170)
	while (p != end) {
		const bool withArrays = size_t(p - data) >= arrayFoundPos;
		uint32_t id = readVarint(p, end);
		uint32_t head = readVarint(p, end);
		const bool idModified = head & 1, fieldIsSame = head & 2, sizeIs1 = head & 4;
		const bool arrayIdxIsZero = withArrays ? bool(head & 8) : true;
		uint32_t pos = head >> (withArrays ? 4 : 3);
		if (idModified) id += lastId;
		uint32_t field = fieldIsSame ? lastField : readVarint(p, end);
		uint32_t arrayIdx = arrayIdxIsZero ? 0 : readVarint(p, end) + 1;
		const uint32_t size = sizeIs1 ? 1 : readVarint(p, end) + 1;
		doc.push_back(id);
		fpos.push_back(Pos(pos, field, arrayIdx));
		const uint32_t firstField = field;
		for (uint32_t i = 1; i < size; ++i) {
			uint32_t next = readVarint(p, end);
			const bool sameField = next & 1;
			if (withArrays) {
				const bool sameArrayIdx = next & 2;
				next >>= 2;
				if (sameField && sameArrayIdx) next += pos;
				if (!sameField) field += readVarint(p, end);
				if (!sameArrayIdx) {
					const uint32_t a = readVarint(p, end);
					arrayIdx = a + (sameField ? arrayIdx : 0);
				}
			} else {
				next >>= 1;
				if (sameField) {
					next += pos;
				} else {
					field += readVarint(p, end);
				}
			}
			pos = next;
			fpos.push_back(Pos(pos, field, arrayIdx));
		}
		posOff.push_back(uint32_t(fpos.size()));
		lastId = id;
		lastField = firstField;
	}
}

GpuFtMerger::GpuFtMerger(size_t numFields, int device) : numFields_(numFields) {
	if (rxgpu_ft_create(uint32_t(numFields), device, &dev_) != RXGPU_OK) throwDevice("GpuFtMerger: device index creation failed");
}

GpuFtMerger::GpuFtMerger(size_t numFields, std::vector<int> devices) : numFields_(numFields) {
	if (devices.empty()) throw std::logic_error("GpuFtMerger: empty device list");
	if (devices.size() == 1) {
		if (rxgpu_ft_create(uint32_t(numFields), devices[0], &dev_) != RXGPU_OK) throwDevice("GpuFtMerger: device index creation failed");
		return;
	}
	if (rxgpu_ft_create_sharded(uint32_t(numFields), uint32_t(devices.size()), devices.data(), &dev_) != RXGPU_OK) {
		throwDevice("GpuFtMerger: sharded device index creation failed");
	}
	sharded_ = true;
}

GpuFtMerger::~GpuFtMerger() {
	if (dev_) rxgpu_ft_destroy(dev_);
}

double GpuFtMerger::ShardImbalance() const noexcept { return sharded_ ? rxgpu_ft_shard_imbalance(dev_) : 1.0; }

void GpuFtMerger::SetDocs(size_t totalDocs, const float* wordsInField, const float* avgWords, const uint8_t* removed) {
	if (rxgpu_ft_set_docs(dev_, totalDocs, wordsInField, avgWords, removed) != RXGPU_OK) throwDevice("SetDocs");
	totalDocs_ = totalDocs;
	words_.assign(wordsInField, wordsInField + totalDocs * numFields_);
}

void GpuFtMerger::SetWord(uint32_t wordId, const FlatPostings& p) {
	if (rxgpu_ft_set_word(dev_, wordId, p.doc.size(), p.doc.data(), p.entOff.data(), p.entField.data(), p.entTf.data(), p.entFirstPos.data()) !=
		RXGPU_OK) {
		throwDevice("SetWord");
	}
}

void GpuFtMerger::ReadStats(uint64_t& postings, double& kernelMs) const {
	if (rxgpu_ft_read_stats(dev_, &postings, &kernelMs) != RXGPU_OK) throwDevice("ReadStats");
}

void GpuFtMerger::ReadPackedStats(double& countMs, double& writeMs, uint64_t& bytesIn, uint64_t& bytesOut) const {
	if (rxgpu_ft_read_packed_stats(dev_, &countMs, &writeMs, &bytesIn, &bytesOut) != RXGPU_OK) throwDevice("ReadPackedStats");
}

void GpuFtMerger::ReadFuseStats(uint64_t& calls, double& kernelMs, double* prepareMs) const {
	if (rxgpu_hybrid_read_stats(dev_, &calls, &kernelMs, prepareMs) != RXGPU_OK) throwDevice("ReadFuseStats");
}

namespace {
struct CallTimer {
	std::atomic<uint64_t>& calls;
	std::atomic<uint64_t>& ns;
	const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	~CallTimer() {
		ns += uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
		++calls;
	}
};
}  // namespace

MergeData GpuFtMerger::Merge(const FtConfig& cfg, const FtDslOpts& termOpts, std::vector<SubtermRef> subterms, const uint8_t* docsExcluded,
							 RankSortType rankSortType) const {
	return mergeImpl(cfg, termOpts, std::move(subterms), docsExcluded, rankSortType, false);
}

MergeData GpuFtMerger::mergeImpl(const FtConfig& cfg, const FtDslOpts& termOpts, std::vector<SubtermRef> subterms, const uint8_t* docsExcluded,
								 RankSortType rankSortType, bool resident) const {
	CallTimer timer{timedCalls_, timedNs_};
	MergeData out;
	if (subterms.empty() || totalDocs_ == 0) return out;   // mergerimpl.h:472-474
	if (cfg.fieldsCfg.size() != numFields_ || termOpts.fieldsOpts.size() != numFields_) throw std::logic_error("GpuFtMerger: field count mismatch");
	// TermResults::SortSubterms (querymergedata.h:62-66): by proc, descending
	std::stable_sort(subterms.begin(), subterms.end(), [](const SubtermRef& l, const SubtermRef& r) { return l.proc > r.proc; });

	std::vector<double> bm25Boost(numFields_), bm25Weight(numFields_), tlBoost(numFields_), tlWeight(numFields_), posBoost(numFields_), posWeight(numFields_);
	std::vector<float> fieldBoost(numFields_);
	std::vector<uint8_t> needSum(numFields_);
	for (size_t f = 0; f < numFields_; ++f) {
		bm25Boost[f] = cfg.fieldsCfg[f].bm25Boost;
		bm25Weight[f] = cfg.fieldsCfg[f].bm25Weight;
		tlBoost[f] = cfg.fieldsCfg[f].termLenBoost;
		tlWeight[f] = cfg.fieldsCfg[f].termLenWeight;
		posBoost[f] = cfg.fieldsCfg[f].positionBoost;
		posWeight[f] = cfg.fieldsCfg[f].positionWeight;
		fieldBoost[f] = termOpts.fieldsOpts[f].boost;
		needSum[f] = termOpts.fieldsOpts[f].needSumRank ? 1 : 0;
	}
	rxgpu_ft_config c{};
	c.bm25_type = cfg.bm25Type == FtConfig::Bm25Type::Rx ? 0 : (cfg.bm25Type == FtConfig::Bm25Type::Classic ? 1 : 2);
	c.bm25_k1 = cfg.bm25k1;
	c.bm25_b = cfg.bm25b;
	c.summation_ranks_by_fields_ratio = cfg.summationRanksByFieldsRatio;
	c.full_match_boost = cfg.fullMatchBoost;
	c.min_rank = cfg.minRank;
	c.merge_limit = cfg.mergeLimit;
	c.num_fields = uint32_t(numFields_);
	c.bm25_boost = bm25Boost.data();
	c.bm25_weight = bm25Weight.data();
	c.term_len_boost = tlBoost.data();
	c.term_len_weight = tlWeight.data();
	c.position_boost = posBoost.data();
	c.position_weight = posWeight.data();
	rxgpu_ft_term_opts o{termOpts.boost, termOpts.termLenBoost, fieldBoost.data(), needSum.data()};

	std::vector<uint32_t> wordIds(subterms.size());
	std::vector<float> procs(subterms.size());
	for (size_t i = 0; i < subterms.size(); ++i) {
		wordIds[i] = subterms[i].wordId;
		procs[i] = subterms[i].proc;
	}
	if (resident) {   // the result stays in HBM for FuseResident
		if (rxgpu_ft_merge_simple_resident(dev_, &c, &o, uint32_t(subterms.size()), wordIds.data(), procs.data(), docsExcluded) != RXGPU_OK) {
			throwDevice("MergeQueryResident");
		}
		return out;
	}
	const size_t cap = cfg.mergeLimit;
	std::vector<uint32_t> doc(cap);
	std::vector<float> proc(cap);
	std::vector<uint8_t> field(cap);
	uint64_t n = 0;
	if (rxgpu_ft_merge_simple_raw(dev_, &c, &o, uint32_t(subterms.size()), wordIds.data(), procs.data(), docsExcluded, doc.data(), proc.data(),
								  field.data(), cap, &n) != RXGPU_OK) {
		throwDevice("Merge");
	}
	out.resize(n);
	for (uint64_t i = 0; i < n; ++i) {
		out[i].id = int32_t(doc[i]);
		out[i].proc = proc[i];
		out[i].field = field[i];
	}
	// addFullMatchBoost(numTerms = 1) — merger.h:100-109 — was applied on the device (ft_replay)
	postProcess(cfg, out, rankSortType);
	return out;
}

namespace {
// postProcessResults — merger.h:111-155 (the same for MergeInfo and MergeInfoAreas: the areas stay where they are, areaIndex travels)
template <typename Vec>
void postProcessImpl(const FtConfig& cfg, Vec& out, RankSortType rankSortType) {
	using Info = typename Vec::value_type;
	float maxProc = 0.0f;
	for (const Info& md : out) maxProc = std::max(maxProc, md.proc);
	const float scalingFactor = float(maxProc > 255 ? 255.0 / double(maxProc) : 1.0);
	const float minProc = float(cfg.minRank);
	size_t passed = out.size();
	while (passed > 0 && out[passed - 1].proc < minProc) passed--;
	for (size_t i = 0; i + 1 < passed; i++) {
		if (out[i].proc < minProc) {
			out[i] = out[passed - 1];
			passed--;
			while (passed > i && out[passed - 1].proc < minProc) passed--;
		}
	}
	out.resize(passed);
	for (Info& md : out) {
		md.normalizedProc = uint8_t(md.proc * scalingFactor);
		md.proc = md.normalizedProc;
	}
	if (rankSortType == RankSortType::RankOnly || rankSortType == RankSortType::IDAndPositions) {
		// the key is one byte: a stable counting sort (descending) gives exactly what a stable comparison sort would, in O(n)
		size_t start[257] = {0};
		for (const Info& md : out) ++start[255 - md.normalizedProc + 1];
		for (int b = 0; b < 256; ++b) start[b + 1] += start[b];
		std::vector<Info> sorted(out.size());
		for (const Info& md : out) sorted[start[255 - md.normalizedProc]++] = md;
		std::copy(sorted.begin(), sorted.end(), out.begin());
	}
}
}  // namespace

void GpuFtMerger::postProcess(const FtConfig& cfg, MergeDataAreas& out, RankSortType rankSortType) const { postProcessImpl(cfg, out, rankSortType); }

void GpuFtMerger::postProcess(const FtConfig& cfg, MergeData& out, RankSortType rankSortType) const { postProcessImpl(cfg, out, rankSortType); }

void GpuFtMerger::SetWord(uint32_t wordId, const PositionPostings& p) {
	if (rxgpu_ft_set_word_positions(dev_, wordId, p.doc.size(), p.doc.data(), p.posOff.data(), p.fpos.data()) != RXGPU_OK) throwDevice("SetWord");
}

void GpuFtMerger::SetWordsPacked(const std::vector<PackedWord>& words, size_t hostDecodeFromBytes) {
	std::vector<uint32_t> ids;
	std::vector<const uint8_t*> data;
	std::vector<uint64_t> len, afp;
	for (const PackedWord& w : words) {
		if (sharded_ || w.len >= hostDecodeFromBytes) {   // (a sharded index cuts every list at its document ranges: decoded here, split by the library)
			PositionPostings pp;
			pp.AppendPacked(w.data, w.len, w.arrayFoundPos);
			SetWord(w.wordId, pp);
			continue;
		}
		ids.push_back(w.wordId);
		data.push_back(w.data);
		len.push_back(w.len);
		afp.push_back(w.arrayFoundPos);
	}
	if (ids.empty()) return;
	// the streams stay where the dictionary keeps them: the library gathers them once, straight into pinned staging memory
	if (rxgpu_ft_set_words_packed_ptrs(dev_, uint32_t(ids.size()), ids.data(), data.data(), len.data(), afp.data()) != RXGPU_OK) throwDevice("SetWordsPacked");
}

void GpuFtMerger::ReadPackedWall(double& wallMs) const {
	if (rxgpu_ft_read_packed_wall(dev_, &wallMs) != RXGPU_OK) throwDevice("ReadPackedWall");
}

void GpuFtMerger::GetWord(uint32_t wordId, PositionPostings& positions, FlatPostings& entries, std::vector<uint32_t>& rangeOff) const {
	uint64_t n = 0, npos = 0, nent = 0;
	uint32_t nRanges = 0;
	if (rxgpu_ft_get_word(dev_, wordId, &n, &npos, &nent, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &nRanges, nullptr) != RXGPU_OK) {
		throwDevice("GetWord");
	}
	positions.doc.assign(n, 0);
	positions.posOff.assign(n + 1, 0);
	positions.fpos.assign(npos, 0);
	entries.doc.assign(n, 0);
	entries.entOff.assign(n + 1, 0);
	entries.entField.assign(nent, 0);
	entries.entTf.assign(nent, 0);
	entries.entFirstPos.assign(nent, 0);
	rangeOff.assign(nRanges, 0);
	if (!n) return;
	if (rxgpu_ft_get_word(dev_, wordId, &n, &npos, &nent, positions.doc.data(), npos ? positions.posOff.data() : nullptr, positions.fpos.data(),
						  entries.entOff.data(), entries.entField.data(), entries.entTf.data(), entries.entFirstPos.data(), &nRanges,
						  rangeOff.data()) != RXGPU_OK) {
		throwDevice("GetWord");
	}
	entries.doc = positions.doc;
}

void GpuFtMerger::ReadTiming(uint64_t& calls, double& totalMs) const {
	calls = timedCalls_.exchange(0);
	totalMs = double(timedNs_.exchange(0)) * 1e-6;
}

MergeData GpuFtMerger::MergeQuery(const FtConfig& cfg, std::vector<QueryTerm> terms, const uint8_t* docsExcluded, RankSortType rankSortType,
								  bool* preselected) const {
	return mergeQueryImpl(cfg, std::move(terms), docsExcluded, rankSortType, preselected, false);
}

bool GpuFtMerger::MergeQueryResident(const FtConfig& cfg, std::vector<QueryTerm> terms, const uint8_t* docsExcluded) const {
	// what Merge() returns empty without touching a posting (mergerimpl.h:472-474; a sub-term-less query merges nothing): no resident result
	if (terms.empty() || (terms.size() == 1 && terms[0].op == OpType::Not) || totalDocs_ == 0) return false;
	size_t subs = 0;
	for (const QueryTerm& t : terms) subs += t.subterms.size();
	if (subs == 0) return false;
	if (terms.size() == 1 && terms[0].subterms.empty()) return false;
	if (terms.front().phraseNum >= 0 && terms.front().op == OpType::Not && terms.front().phraseNum == terms.back().phraseNum) return false;   // one NOT phrase: Empty()
	(void)mergeQueryImpl(cfg, std::move(terms), docsExcluded, RankSortType::RankAndID, nullptr, true);
	return true;
}

bool GpuFtMerger::MergeQueryResident(const FtConfig& cfg, std::vector<QueryTerm> terms, QuerySynonyms synonyms, const uint8_t* docsExcluded) const {
	if (synonyms.Empty()) return MergeQueryResident(cfg, std::move(terms), docsExcluded);
	// QueryMergeData::Empty() looks at the query parts only: the same early answers as above
	if (terms.empty() || (terms.size() == 1 && terms[0].op == OpType::Not) || totalDocs_ == 0) return false;
	if (terms.front().phraseNum >= 0 && terms.front().op == OpType::Not && terms.front().phraseNum == terms.back().phraseNum) return false;
	(void)mergeQueryImpl(cfg, std::move(terms), docsExcluded, RankSortType::RankAndID, nullptr, true, &synonyms);
	return true;
}

namespace {
rxgpu_hybrid_params toAbi(const HybridFuseParams& hp) {
	rxgpu_hybrid_params p{};
	p.kind = hp.linear ? 1 : 0;
	p.is_union = hp.isUnion ? 1 : 0;
	p.desc = hp.desc ? 1 : 0;
	for (int i = 0; i < 5; ++i) p.params[i] = hp.params[i];
	return p;
}
}  // namespace

void GpuFtMerger::PrepareResident(const FtConfig& cfg, const HybridFuseParams& hp, int metric, const void* dRowOfDoc) const {
	const rxgpu_hybrid_params p = toAbi(hp);
	if (rxgpu_hybrid_prepare_resident(dev_, cfg.minRank, &p, metric, dRowOfDoc) != RXGPU_OK) throwDevice("PrepareResident");
}

HybridFused GpuFtMerger::FuseResident(const FtConfig& cfg, const HybridFuseParams& hp, int metric, const void* dKnnDist, const void* dKnnRow,
									  const void* dKnnCount, uint32_t knnEntries, uint32_t k, void* knnStream, const void* dRowOfDoc,
									  const void* dRowIdOfRow) const {
	HybridFused out;
	const size_t cap = size_t(cfg.mergeLimit) + k;
	out.ids.resize(cap);
	out.ranks.resize(cap);
	const rxgpu_hybrid_params p = toAbi(hp);
	uint64_t n = 0;
	uint32_t flags = 0;
	if (rxgpu_hybrid_fuse_resident(dev_, cfg.minRank, &p, metric, dKnnDist, dKnnRow, dKnnCount, knnEntries, k, knnStream, dRowOfDoc, dRowIdOfRow,
								   out.ids.data(), out.ranks.data(), cap, &n, &flags) != RXGPU_OK) {
		throwDevice("FuseResident");
	}
	out.ids.resize(n);
	out.ranks.resize(n);
	out.knnBoundaryTie = (flags & 1u) != 0;
	return out;
}

MergeData GpuFtMerger::MergeQuery(const FtConfig& cfg, std::vector<QueryTerm> terms, QuerySynonyms synonyms, const uint8_t* docsExcluded,
								  RankSortType rankSortType, bool* preselected) const {
	return mergeQueryImpl(cfg, std::move(terms), docsExcluded, rankSortType, preselected, false, synonyms.Empty() ? nullptr : &synonyms);
}

MergeData GpuFtMerger::mergeQueryImpl(const FtConfig& cfg, std::vector<QueryTerm> terms, const uint8_t* docsExcluded, RankSortType rankSortType,
									  bool* preselected, bool resident, QuerySynonyms* synonyms) const {
	if (!synonyms && terms.size() == 1 && terms[0].op != OpType::Not && terms[0].phraseNum < 0 && totalDocs_ != 0) {   // Simple(): timed by Merge
		if (preselected) *preselected = false;
		return mergeImpl(cfg, terms[0].opts, std::move(terms[0].subterms), docsExcluded, rankSortType, resident);
	}
	CallTimer timer{timedCalls_, timedNs_};
	if (preselected) *preselected = false;
	MergeData out;
	// QueryMergeData::Empty() (querymergedata.h:208) / mergerimpl.h:472-474
	if (terms.empty() || (terms.size() == 1 && terms[0].op == OpType::Not) || totalDocs_ == 0) return out;
	bool anyPhrase = false;
	for (const QueryTerm& t : terms) anyPhrase = anyPhrase || t.phraseNum >= 0;
	if (terms.size() == 1 && !anyPhrase && !synonyms) return mergeImpl(cfg, terms[0].opts, std::move(terms[0].subterms), docsExcluded, rankSortType, resident);   // Simple()
	if (cfg.fieldsCfg.size() != numFields_) throw std::logic_error("GpuFtMerger: field count mismatch");

	const size_t npartTerms = terms.size();
	std::vector<uint32_t> synTermOff{0};
	if (synonyms) {   // the synonyms' terms behind the parts' terms
		for (auto& syn : synonyms->synonyms) {
			for (auto& t : syn) terms.push_back(std::move(t));
			synTermOff.push_back(uint32_t(terms.size() - npartTerms));
		}
	}
	const size_t nt = terms.size();
	std::vector<double> bm25Boost(numFields_), bm25Weight(numFields_), tlBoost(numFields_), tlWeight(numFields_), posBoost(numFields_), posWeight(numFields_);
	for (size_t f = 0; f < numFields_; ++f) {
		bm25Boost[f] = cfg.fieldsCfg[f].bm25Boost;
		bm25Weight[f] = cfg.fieldsCfg[f].bm25Weight;
		tlBoost[f] = cfg.fieldsCfg[f].termLenBoost;
		tlWeight[f] = cfg.fieldsCfg[f].termLenWeight;
		posBoost[f] = cfg.fieldsCfg[f].positionBoost;
		posWeight[f] = cfg.fieldsCfg[f].positionWeight;
	}
	rxgpu_ft_config c{};
	c.bm25_type = cfg.bm25Type == FtConfig::Bm25Type::Rx ? 0 : (cfg.bm25Type == FtConfig::Bm25Type::Classic ? 1 : 2);
	c.bm25_k1 = cfg.bm25k1;
	c.bm25_b = cfg.bm25b;
	c.summation_ranks_by_fields_ratio = cfg.summationRanksByFieldsRatio;
	c.full_match_boost = cfg.fullMatchBoost;
	c.min_rank = cfg.minRank;
	c.merge_limit = cfg.mergeLimit;
	c.num_fields = uint32_t(numFields_);
	c.bm25_boost = bm25Boost.data();
	c.bm25_weight = bm25Weight.data();
	c.term_len_boost = tlBoost.data();
	c.term_len_weight = tlWeight.data();
	c.position_boost = posBoost.data();
	c.position_weight = posWeight.data();
	c.distance_boost = cfg.distanceBoost;
	c.distance_weight = cfg.distanceWeight;

	std::vector<int32_t> ops(nt), phraseNum(nt), distance(nt);
	std::vector<float> fieldBoost(nt * numFields_);
	std::vector<uint8_t> needSum(nt * numFields_);
	std::vector<rxgpu_ft_term_opts> opts(nt);
	std::vector<uint32_t> subOff(nt + 1, 0), wordIds;
	std::vector<float> procs;
	std::vector<uint8_t> suppressed;
	for (size_t t = 0; t < nt; ++t) {
		QueryTerm& qt = terms[t];
		if (qt.opts.fieldsOpts.size() != numFields_) throw std::logic_error("GpuFtMerger: field count mismatch");
		ops[t] = int32_t(qt.op);
		phraseNum[t] = qt.phraseNum;
		distance[t] = qt.distance;
		for (size_t f = 0; f < numFields_; ++f) {
			fieldBoost[t * numFields_ + f] = qt.opts.fieldsOpts[f].boost;
			needSum[t * numFields_ + f] = qt.opts.fieldsOpts[f].needSumRank ? 1 : 0;
		}
		opts[t] = rxgpu_ft_term_opts{qt.opts.boost, qt.opts.termLenBoost, fieldBoost.data() + t * numFields_, needSum.data() + t * numFields_};
		// QueryMergeData::SortSubterms (querymergedata.h:196-206)
		std::stable_sort(qt.subterms.begin(), qt.subterms.end(), [](const SubtermRef& l, const SubtermRef& r) { return l.proc > r.proc; });
		for (const SubtermRef& sr : qt.subterms) {
			wordIds.push_back(sr.wordId);
			procs.push_back(sr.proc);
			suppressed.push_back(sr.suppressed ? 1 : 0);
		}
		subOff[t + 1] = uint32_t(wordIds.size());
	}
	if (synonyms) {
		// parts as the engine forms them: a plain term, or the consecutive terms of one phrase number
		uint32_t nparts = 0;
		for (size_t t = 0; t < npartTerms; ++t) {
			if (terms[t].phraseNum < 0 || t == 0 || terms[t - 1].phraseNum != terms[t].phraseNum) ++nparts;
		}
		std::vector<uint32_t> partSynOff(nparts + 1, 0), partSyn;
		for (uint32_t pi = 0; pi < nparts; ++pi) {
			if (pi < synonyms->partSynonyms.size()) {
				for (uint32_t id : synonyms->partSynonyms[pi]) {
					if (id >= synonyms->synonyms.size()) throw std::logic_error("GpuFtMerger: synonym id out of range");
					partSyn.push_back(id);
				}
			}
			partSynOff[pi + 1] = uint32_t(partSyn.size());
		}
		rxgpu_ft_query q{};
		q.nterms = uint32_t(npartTerms);
		q.nsyn_terms = uint32_t(nt - npartTerms);
		q.ops = ops.data();
		q.opts = opts.data();
		q.phrase_num = phraseNum.data();
		q.distance = distance.data();
		q.sub_off = subOff.data();
		q.word_ids = wordIds.data();
		q.procs = procs.data();
		q.suppressed = suppressed.data();
		q.nsyn = uint32_t(synonyms->synonyms.size());
		q.syn_term_off = synTermOff.data();
		q.part_syn_off = partSynOff.data();
		q.part_syn = partSyn.data();
		if (resident) {   // left in HBM for FuseResident; the removed documents keep their mark there
			int32_t enqueued = 0;
			if (rxgpu_ft_merge_query2_resident(dev_, &c, &q, docsExcluded, &enqueued) != RXGPU_OK) throwDevice("MergeQueryResident (synonyms)");
			return out;
		}
		const size_t cap = cfg.mergeLimit;
		std::vector<uint32_t> doc(cap);
		std::vector<float> proc(cap);
		std::vector<uint8_t> field(cap);
		std::vector<uint16_t> termsCounter(cap);
		uint64_t n = 0;
		int32_t pre = 0;
		if (rxgpu_ft_merge_query2_raw(dev_, &c, &q, docsExcluded, doc.data(), proc.data(), field.data(), termsCounter.data(), cap, &n, &pre) != RXGPU_OK) {
			throwDevice("MergeQuery (synonyms)");
		}
		if (preselected) *preselected = pre != 0;
		out.resize(n);
		for (uint64_t i = 0; i < n; ++i) {
			out[i].id = int32_t(doc[i]);
			out[i].proc = proc[i];
			out[i].field = field[i];
		}
		postProcess(cfg, out, rankSortType);
		return out;
	}
	if (resident && anyPhrase) {
		int32_t enqueued = 0;
		if (rxgpu_ft_merge_query_resident(dev_, &c, uint32_t(nt), ops.data(), opts.data(), phraseNum.data(), distance.data(), subOff.data(), wordIds.data(),
										  procs.data(), docsExcluded, &enqueued) != RXGPU_OK) {
			throwDevice("MergeQueryResident");
		}
		return out;
	}
	if (resident) {
		if (rxgpu_ft_merge_terms_resident(dev_, &c, uint32_t(nt), ops.data(), opts.data(), subOff.data(), wordIds.data(), procs.data(), docsExcluded) !=
			RXGPU_OK) {
			throwDevice("MergeQueryResident");
		}
		return out;
	}
	const size_t cap = cfg.mergeLimit;
	std::vector<uint32_t> doc(cap);
	std::vector<float> proc(cap);
	std::vector<uint8_t> field(cap);
	std::vector<uint16_t> termsCounter(cap);
	uint64_t n = 0;
	int32_t pre = 0;
	const int rc = anyPhrase ? rxgpu_ft_merge_query_raw(dev_, &c, uint32_t(nt), ops.data(), opts.data(), phraseNum.data(), distance.data(), subOff.data(),
														wordIds.data(), procs.data(), docsExcluded, doc.data(), proc.data(), field.data(),
														termsCounter.data(), cap, &n, &pre)
							 : rxgpu_ft_merge_terms_raw(dev_, &c, uint32_t(nt), ops.data(), opts.data(), subOff.data(), wordIds.data(), procs.data(), docsExcluded,
														doc.data(), proc.data(), field.data(), termsCounter.data(), cap, &n, &pre);
	if (rc != RXGPU_OK) throwDevice("MergeQuery");
	if (preselected) *preselected = pre != 0;
	out.resize(n);
	// canBeBoostedByFullMatch / addFullMatchBoost(QueryLength) (mergerimpl.h:527-531, merger.h:100-109) were applied on the device (ft_replay)
	for (uint64_t i = 0; i < n; ++i) {
		out[i].id = int32_t(doc[i]);
		out[i].proc = proc[i];
		out[i].field = field[i];
	}
	postProcess(cfg, out, rankSortType);
	return out;
}

std::vector<MergeData> GpuFtMerger::MergeQueryBatch(const FtConfig& cfg, std::vector<std::vector<QueryTerm>> queries, const std::vector<const uint8_t*>& docsExcluded,
													RankSortType rankSortType, std::vector<uint8_t>* preselected) const {
	CallTimer timer{timedCalls_, timedNs_};
	const size_t nq = queries.size();
	std::vector<MergeData> out(nq);
	if (preselected) preselected->assign(nq, 0);
	if (!nq || totalDocs_ == 0) return out;
	if (cfg.fieldsCfg.size() != numFields_) throw std::logic_error("GpuFtMerger: field count mismatch");
	if (!docsExcluded.empty() && docsExcluded.size() != nq) throw std::logic_error("GpuFtMerger::MergeQueryBatch: one docsExcluded per query (or none)");
	// (over a device list rxgpu_ft_merge_batch_raw runs the merges one after the other: every shard's handle runs one launch train and its
	// exchanges at a time; each result is the single sharded merge's, i.e. the single index's)
	std::vector<double> bm25Boost(numFields_), bm25Weight(numFields_), tlBoost(numFields_), tlWeight(numFields_), posBoost(numFields_), posWeight(numFields_);
	for (size_t f = 0; f < numFields_; ++f) {
		bm25Boost[f] = cfg.fieldsCfg[f].bm25Boost;
		bm25Weight[f] = cfg.fieldsCfg[f].bm25Weight;
		tlBoost[f] = cfg.fieldsCfg[f].termLenBoost;
		tlWeight[f] = cfg.fieldsCfg[f].termLenWeight;
		posBoost[f] = cfg.fieldsCfg[f].positionBoost;
		posWeight[f] = cfg.fieldsCfg[f].positionWeight;
	}
	rxgpu_ft_config c{};
	c.bm25_type = cfg.bm25Type == FtConfig::Bm25Type::Rx ? 0 : (cfg.bm25Type == FtConfig::Bm25Type::Classic ? 1 : 2);
	c.bm25_k1 = cfg.bm25k1;
	c.bm25_b = cfg.bm25b;
	c.summation_ranks_by_fields_ratio = cfg.summationRanksByFieldsRatio;
	c.full_match_boost = cfg.fullMatchBoost;
	c.min_rank = cfg.minRank;
	c.merge_limit = cfg.mergeLimit;
	c.num_fields = uint32_t(numFields_);
	c.bm25_boost = bm25Boost.data();
	c.bm25_weight = bm25Weight.data();
	c.term_len_boost = tlBoost.data();
	c.term_len_weight = tlWeight.data();
	c.position_boost = posBoost.data();
	c.position_weight = posWeight.data();
	c.distance_boost = cfg.distanceBoost;
	c.distance_weight = cfg.distanceWeight;

	// one query's arrays as the C-ABI wants them; the vectors own what rxgpu_ft_query points at
	struct AbiQuery {
		std::vector<int32_t> ops, phraseNum, distance;
		std::vector<float> fieldBoost, procs;
		std::vector<uint8_t> needSum;
		std::vector<rxgpu_ft_term_opts> opts;
		std::vector<uint32_t> subOff, wordIds;
	};
	std::vector<AbiQuery> abi(nq);
	std::vector<rxgpu_ft_query> qs(nq);
	for (size_t i = 0; i < nq; ++i) {
		std::vector<QueryTerm>& terms = queries[i];
		AbiQuery& a = abi[i];
		const size_t nt = terms.size();
		a.ops.resize(nt);
		a.phraseNum.resize(nt);
		a.distance.resize(nt);
		a.fieldBoost.resize(nt * numFields_);
		a.needSum.resize(nt * numFields_);
		a.opts.resize(nt);
		a.subOff.assign(nt + 1, 0);
		for (size_t t = 0; t < nt; ++t) {
			QueryTerm& qt = terms[t];
			if (qt.opts.fieldsOpts.size() != numFields_) throw std::logic_error("GpuFtMerger: field count mismatch");
			a.ops[t] = int32_t(qt.op);
			a.phraseNum[t] = qt.phraseNum;
			a.distance[t] = qt.distance;
			for (size_t f = 0; f < numFields_; ++f) {
				a.fieldBoost[t * numFields_ + f] = qt.opts.fieldsOpts[f].boost;
				a.needSum[t * numFields_ + f] = qt.opts.fieldsOpts[f].needSumRank ? 1 : 0;
			}
			a.opts[t] = rxgpu_ft_term_opts{qt.opts.boost, qt.opts.termLenBoost, a.fieldBoost.data() + t * numFields_, a.needSum.data() + t * numFields_};
			// QueryMergeData::SortSubterms (querymergedata.h:196-206)
			std::stable_sort(qt.subterms.begin(), qt.subterms.end(), [](const SubtermRef& l, const SubtermRef& r) { return l.proc > r.proc; });
			for (const SubtermRef& sr : qt.subterms) {
				a.wordIds.push_back(sr.wordId);
				a.procs.push_back(sr.proc);
			}
			a.subOff[t + 1] = uint32_t(a.wordIds.size());
		}
		rxgpu_ft_query& q = qs[i];
		q = rxgpu_ft_query{};
		q.nterms = uint32_t(nt);
		q.ops = a.ops.data();
		q.opts = a.opts.data();
		q.phrase_num = a.phraseNum.data();
		q.distance = a.distance.data();
		q.sub_off = a.subOff.data();
		q.word_ids = a.wordIds.data();
		q.procs = a.procs.data();
	}
	const size_t cap = cfg.mergeLimit;
	std::vector<uint32_t> doc(nq * cap);
	std::vector<float> proc(nq * cap);
	std::vector<uint8_t> field(nq * cap);
	std::vector<uint16_t> termsCounter(nq * cap);
	std::vector<uint32_t*> pDoc(nq);
	std::vector<float*> pProc(nq);
	std::vector<uint8_t*> pField(nq);
	std::vector<uint16_t*> pTc(nq);
	for (size_t i = 0; i < nq; ++i) {
		pDoc[i] = doc.data() + i * cap;
		pProc[i] = proc.data() + i * cap;
		pField[i] = field.data() + i * cap;
		pTc[i] = termsCounter.data() + i * cap;
	}
	std::vector<uint64_t> n(nq, 0);
	std::vector<int32_t> pre(nq, 0);
	if (rxgpu_ft_merge_batch_raw(dev_, &c, uint32_t(nq), qs.data(), docsExcluded.empty() ? nullptr : docsExcluded.data(), pDoc.data(), pProc.data(), pField.data(),
								 pTc.data(), cap, n.data(), pre.data()) != RXGPU_OK) {
		throwDevice("MergeQueryBatch");
	}
	for (size_t i = 0; i < nq; ++i) {
		if (preselected) (*preselected)[i] = pre[i] ? 1 : 0;
		MergeData& md = out[i];
		md.resize(n[i]);
		for (uint64_t j = 0; j < n[i]; ++j) {
			md[j].id = int32_t(pDoc[i][j]);
			md[j].proc = pProc[i][j];
			md[j].field = pField[i][j];
		}
		postProcess(cfg, md, rankSortType);
	}
	return out;
}

MergeDataAreas GpuFtMerger::MergeQueryAreas(const FtConfig& cfg, std::vector<QueryTerm> terms, const uint8_t* docsExcluded, RankSortType rankSortType,
											int maxAreasInDoc, bool* preselected) const {
	CallTimer timer{timedCalls_, timedNs_};
	if (preselected) *preselected = false;
	MergeDataAreas out;
	if (terms.empty() || (terms.size() == 1 && terms[0].op == OpType::Not) || totalDocs_ == 0) return out;   // Empty() / mergerimpl.h:472-474
	if (cfg.fieldsCfg.size() != numFields_) throw std::logic_error("GpuFtMerger: field count mismatch");
	bool anyPhrase = false;
	for (const QueryTerm& t : terms) anyPhrase = anyPhrase || t.phraseNum >= 0;
	if (!SupportsAreas(terms.size(), anyPhrase, false, maxAreasInDoc)) throw std::logic_error("GpuFtMerger::MergeQueryAreas: this query's areas are built by the CPU merger (SupportsAreas)");
	const size_t nt = terms.size();
	std::vector<double> bm25Boost(numFields_), bm25Weight(numFields_), tlBoost(numFields_), tlWeight(numFields_), posBoost(numFields_), posWeight(numFields_);
	for (size_t f = 0; f < numFields_; ++f) {
		bm25Boost[f] = cfg.fieldsCfg[f].bm25Boost;
		bm25Weight[f] = cfg.fieldsCfg[f].bm25Weight;
		tlBoost[f] = cfg.fieldsCfg[f].termLenBoost;
		tlWeight[f] = cfg.fieldsCfg[f].termLenWeight;
		posBoost[f] = cfg.fieldsCfg[f].positionBoost;
		posWeight[f] = cfg.fieldsCfg[f].positionWeight;
	}
	rxgpu_ft_config c{};
	c.bm25_type = cfg.bm25Type == FtConfig::Bm25Type::Rx ? 0 : (cfg.bm25Type == FtConfig::Bm25Type::Classic ? 1 : 2);
	c.bm25_k1 = cfg.bm25k1;
	c.bm25_b = cfg.bm25b;
	c.summation_ranks_by_fields_ratio = cfg.summationRanksByFieldsRatio;
	c.full_match_boost = cfg.fullMatchBoost;
	c.min_rank = cfg.minRank;
	c.merge_limit = cfg.mergeLimit;
	c.num_fields = uint32_t(numFields_);
	c.bm25_boost = bm25Boost.data();
	c.bm25_weight = bm25Weight.data();
	c.term_len_boost = tlBoost.data();
	c.term_len_weight = tlWeight.data();
	c.position_boost = posBoost.data();
	c.position_weight = posWeight.data();
	c.distance_boost = cfg.distanceBoost;
	c.distance_weight = cfg.distanceWeight;
	std::vector<int32_t> ops(nt);
	std::vector<float> fieldBoost(nt * numFields_), procs;
	std::vector<uint8_t> needSum(nt * numFields_);
	std::vector<rxgpu_ft_term_opts> opts(nt);
	std::vector<uint32_t> subOff(nt + 1, 0), wordIds;
	for (size_t t = 0; t < nt; ++t) {
		QueryTerm& qt = terms[t];
		if (qt.opts.fieldsOpts.size() != numFields_) throw std::logic_error("GpuFtMerger: field count mismatch");
		ops[t] = int32_t(qt.op);
		for (size_t f = 0; f < numFields_; ++f) {
			fieldBoost[t * numFields_ + f] = qt.opts.fieldsOpts[f].boost;
			needSum[t * numFields_ + f] = qt.opts.fieldsOpts[f].needSumRank ? 1 : 0;
		}
		opts[t] = rxgpu_ft_term_opts{qt.opts.boost, qt.opts.termLenBoost, fieldBoost.data() + t * numFields_, needSum.data() + t * numFields_};
		std::stable_sort(qt.subterms.begin(), qt.subterms.end(), [](const SubtermRef& l, const SubtermRef& r) { return l.proc > r.proc; });   // SortSubterms
		for (const SubtermRef& sr : qt.subterms) {
			wordIds.push_back(sr.wordId);
			procs.push_back(sr.proc);
		}
		subOff[t + 1] = uint32_t(wordIds.size());
	}
	rxgpu_ft_query q{};
	q.nterms = uint32_t(nt);
	q.ops = ops.data();
	q.opts = opts.data();
	q.sub_off = subOff.data();
	q.word_ids = wordIds.data();
	q.procs = procs.data();
	const size_t cap = cfg.mergeLimit, nf = numFields_, maxA = size_t(maxAreasInDoc);
	std::vector<uint32_t> doc(cap), areaCnt(cap * nf), areas(cap * nf * maxA * 3);
	std::vector<float> proc(cap);
	std::vector<uint8_t> field(cap);
	std::vector<uint16_t> termsCounter(cap);
	uint64_t n = 0;
	int32_t pre = 0;
	if (rxgpu_ft_merge_query_areas_raw(dev_, &c, &q, docsExcluded, uint32_t(maxAreasInDoc), doc.data(), proc.data(), field.data(), termsCounter.data(), cap, &n, &pre,
									   areaCnt.data(), areas.data()) != RXGPU_OK) {
		throwDevice("MergeQueryAreas");
	}
	if (preselected) *preselected = pre != 0;
	out.resize(n);
	out.vectorAreas.resize(n);
	for (uint64_t i = 0; i < n; ++i) {
		out[i].id = int32_t(doc[i]);
		out[i].proc = proc[i];
		out[i].field = field[i];
		out[i].areaIndex = uint32_t(i);
		auto& fields = out.vectorAreas[i];
		fields.resize(nf);   // AreasInDocument::ReserveField(fieldSize_) (merger.h:165-166)
		for (size_t f = 0; f < nf; ++f) {
			const uint32_t cnt = areaCnt[i * nf + f];
			const uint32_t* a = areas.data() + (i * nf + f) * maxA * 3;
			fields[f].data.resize(cnt);
			for (uint32_t j = 0; j < cnt; ++j) fields[f].data[j] = Area{a[j * 3], a[j * 3 + 1], a[j * 3 + 2]};
		}
	}
	postProcess(cfg, out, rankSortType);
	return out;
}

}  // namespace rxgpu::host
