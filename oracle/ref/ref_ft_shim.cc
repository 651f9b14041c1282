// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// Flat C wrapper around the REAL ft_fast merger: reindexer::ft::Merger<IdRelVec, ft::MergeData, uint32_t>::Merge<Bm25Rx>
// (cpp_src/core/ft/ft_fast/merger.h:36-57, mergerimpl.h:466-566), instantiated in place from /root/reference/cpp_src with a
// duck-typed DocsStatsGetter exactly like IndexText's (indextext.h:245-258).  Output: oracle/_ref/libref_ft.so.
// The reference TUs it needs (ftconfig.cc, rdxcontext.cc, stop words, string tools, idrelset.cc, errors.cc) are compiled where
// they lie; symbols that those TUs reference but this path never calls (JSON builders, activity context ...) are satisfied by
// trap stubs GENERATED at build time from `nm -u` (oracle/Makefile) — no reference source is copied or restated.
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "tools/float_comparison.h"
#define private public   /* test infrastructure only: PackedIdRelVec::data_ / arrayFoundPos_ have no accessors */
#include "core/ft/idrelset.h"
#undef private
#include "core/ft/config/ftconfig.h"
#include "core/ft/ft_fast/mergerimpl.h"

using namespace reindexer;

namespace {

struct Stats {
	const float* words;
	size_t nf;
	const float* avg;
	const uint8_t* removed;
	bool DocRemoved(size_t d) const { return removed && removed[d]; }
	float NumWordsInField(size_t d, unsigned f) const { return words[d * nf + f]; }
	float AvgWordsCount(unsigned f) const { return avg[f]; }
};

struct FtRef {
	size_t nf = 1;
	size_t totalDocs = 0;
	std::vector<float> words, avg;
	std::vector<uint8_t> removed;
	bool hasRemoved = false;
	std::vector<std::unique_ptr<IdRelVec>> postings;   // word id -> posting list
	FTConfig cfg{1};
	explicit FtRef(size_t fields) : nf(fields), cfg(fields) {}
};

}  // namespace

extern "C" {

void* ref_ft_create(size_t numFields) { return new FtRef(numFields); }
void ref_ft_destroy(void* h) { delete static_cast<FtRef*>(h); }

void ref_ft_set_docs(void* h, size_t totalDocs, const float* words, const float* avg, const uint8_t* removed) {
	auto* f = static_cast<FtRef*>(h);
	f->totalDocs = totalDocs;
	f->words.assign(words, words + totalDocs * f->nf);
	f->avg.assign(avg, avg + f->nf);
	f->hasRemoved = removed != nullptr;
	if (removed) f->removed.assign(removed, removed + totalDocs);
}

// postings of one dictionary word: doc[i] with positions [posOff[i], posOff[i+1]) given as (field, pos, arrayIdx) triplets
void ref_ft_set_word(void* h, uint32_t wordId, size_t n, const uint32_t* doc, const uint32_t* posOff, const uint32_t* posField,
					 const uint32_t* posPos, const uint32_t* posArrayIdx) {
	auto* f = static_cast<FtRef*>(h);
	if (f->postings.size() <= wordId) f->postings.resize(wordId + 1);
	auto vec = std::make_unique<IdRelVec>();
	vec->reserve(n);
	for (size_t i = 0; i < n; ++i) {
		IdRelType rel(doc[i]);
		for (uint32_t j = posOff[i]; j < posOff[i + 1]; ++j) rel.Add(posPos[j], posField[j], posArrayIdx ? posArrayIdx[j] : 0);
		vec->emplace_back(std::move(rel));
	}
	f->postings[wordId] = std::move(vec);
}

// cfgD: k1, b, summationRanksByFieldsRatio, fullMatchBoost, distanceBoost, distanceWeight ; cfgI: minRank, mergeLimit
// fieldCfg [nf][6]: bm25Boost, bm25Weight, termLenBoost, termLenWeight, positionBoost, positionWeight
void ref_ft_set_config(void* h, const double* cfgD, const int* cfgI, const double* fieldCfg) {
	auto* f = static_cast<FtRef*>(h);
	f->cfg.bm25Config.bm25k1 = cfgD[0];
	f->cfg.bm25Config.bm25b = cfgD[1];
	f->cfg.summationRanksByFieldsRatio = cfgD[2];
	f->cfg.fullMatchBoost = cfgD[3];
	f->cfg.distanceBoost = cfgD[4];
	f->cfg.distanceWeight = cfgD[5];
	f->cfg.minRank = cfgI[0];
	f->cfg.mergeLimit = uint32_t(cfgI[1]);
	f->cfg.fieldsCfg.resize(f->nf);
	for (size_t i = 0; i < f->nf; ++i) {
		auto& fc = f->cfg.fieldsCfg[i];
		fc.bm25Boost = fieldCfg[i * 6 + 0];
		fc.bm25Weight = fieldCfg[i * 6 + 1];
		fc.termLenBoost = fieldCfg[i * 6 + 2];
		fc.termLenWeight = fieldCfg[i * 6 + 3];
		fc.positionBoost = fieldCfg[i * 6 + 4];
		fc.positionWeight = fieldCfg[i * 6 + 5];
	}
}

// FTConfig::Bm25Config::bm25Type: 0 = rx, 1 = classic, 2 = wordCount (ftconfig.h:200-203)
void ref_ft_set_bm25_type(void* h, int type) {
	using T = FTConfig::Bm25Config::Bm25Type;
	static_cast<FtRef*>(h)->cfg.bm25Config.bm25Type = type == 1 ? T::classic : type == 2 ? T::wordCount : T::rx;
}

// Query = nTerms terms.  Per term t: op (1 OR, 2 AND, 3 NOT), boost, termLenBoost, fieldBoost[nf], needSum[nf], and the sub-term
// slice [subOff[t], subOff[t+1]) of (wordId, proc).  excluded: docsExcluded bitmap (bytes) or null.
// rankSortType: 0 RankOnly, 1 RankAndID, 3 IDOnly, 4 IDAndPositions.  Returns the result count (<= cap written).
// phraseNum / distance: FtDslOpts::phraseNum (-1: a plain term) and FtDslOpts::distance per term, or null.  The query parts are put
// together the way Selector::Process does (selecterimpl.h:482-572): consecutive terms with the same phraseNum >= 0 become one PhraseResults.
// ... plus multi-word synonyms: the per-term arrays hold nTerms + nSynTerms entries, synonym s owns the terms nTerms + synTermOff[s] ..
// nTerms + synTermOff[s + 1]; partSynOff [parts + 1] / partSyn: PhraseOrTerm::SynonymsIds of every query part.  The selecter's
// SupressDuplicatesInSynonyms (selecterimpl.h:606) is called before the merge, like Selector::Process does.
long ref_ft_merge_full(void* h, size_t nTerms, size_t nSynTerms, const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost,
					   const uint8_t* needSum, const int* phraseNum, const int* distance, const uint32_t* subOff, const uint32_t* subWord,
					   const float* subProc, size_t nSyn, const uint32_t* synTermOff, const uint32_t* partSynOff, const uint32_t* partSyn,
					   const uint8_t* excluded, int rankSortType, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm, size_t cap) {
	auto* f = static_cast<FtRef*>(h);
	try {
		ft::QueryMergeData<IdRelVec> q;
		auto makeTerm = [&](size_t t, int phrase, int dist) {
			FtDslOpts o;
			o.phraseNum = phrase;
			o.distance = dist;
			o.op = OpType(ops[t]);
			o.boost = boosts[t];
			o.termLenBoost = termLenBoosts[t];
			o.fieldsOpts.resize(f->nf);
			for (size_t i = 0; i < f->nf; ++i) {
				o.fieldsOpts[i].boost = fieldBoost[t * f->nf + i];
				o.fieldsOpts[i].needSumRank = needSum[t * f->nf + i] != 0;
			}
			ft::TermResults<IdRelVec> tr{FtDSLEntry(std::wstring(L"t") + std::to_wstring(t), o)};
			for (uint32_t s = subOff[t]; s < subOff[t + 1]; ++s) {
				WordIdType wid;
				wid.data = 0;
				wid.SetID(int32_t(subWord[s]));
				tr.AddSubterm(*f->postings.at(subWord[s]), std::string_view("w"), wid, subProc[s]);
			}
			return tr;
		};
		int curPhraseNum = -1;
		ft::PhraseResults<IdRelVec> nextPhrase;
		for (size_t t = 0; t < nTerms; ++t) {
			FtDslOpts o;
			o.phraseNum = phraseNum ? phraseNum[t] : -1;
			o.distance = distance ? distance[t] : 1;
			const bool phraseTerm = o.phraseNum != -1;
			if (!phraseTerm && nextPhrase.NumTerms()) {
				q.queryParts.emplace_back(std::move(nextPhrase));
				nextPhrase.clear();
			}
			o.op = OpType(ops[t]);
			o.boost = boosts[t];
			o.termLenBoost = termLenBoosts[t];
			o.fieldsOpts.resize(f->nf);
			for (size_t i = 0; i < f->nf; ++i) {
				o.fieldsOpts[i].boost = fieldBoost[t * f->nf + i];
				o.fieldsOpts[i].needSumRank = needSum[t * f->nf + i] != 0;
			}
			ft::TermResults<IdRelVec> tr{FtDSLEntry(std::wstring(L"t") + std::to_wstring(t), o)};
			for (uint32_t s = subOff[t]; s < subOff[t + 1]; ++s) {
				WordIdType wid;
				wid.data = 0;
				wid.SetID(int32_t(subWord[s]));
				tr.AddSubterm(*f->postings.at(subWord[s]), std::string_view("w"), wid, subProc[s]);
			}
			q.totalORVids += tr.MaxVDocs();   // selecterimpl.h:546: every term, whatever its operator
			if (phraseTerm) {
				if (nextPhrase.NumTerms() && curPhraseNum != o.phraseNum) {
					q.queryParts.emplace_back(std::move(nextPhrase));
					nextPhrase.clear();
				}
				curPhraseNum = o.phraseNum;
				nextPhrase.Add(std::move(tr));
			} else {
				q.queryParts.emplace_back(std::move(tr));
			}
		}
		if (nextPhrase.NumTerms()) {
			q.queryParts.emplace_back(std::move(nextPhrase));
			nextPhrase.clear();
		}
		for (size_t sy = 0; sy < nSyn; ++sy) {
			ft::Synonym<IdRelVec> syn;
			for (uint32_t k = synTermOff[sy]; k < synTermOff[sy + 1]; ++k) {
				auto tr = makeTerm(nTerms + k, -1, 1);
				q.totalORVids += tr.MaxVDocs();   // selecterimpl.h:443, 462, 595
				syn.AddTerm(std::move(tr));
			}
			q.synonyms.emplace_back(std::move(syn));
		}
		if (nSyn) {
			for (size_t pi = 0; pi < q.queryParts.size(); ++pi) {
				for (uint32_t k = partSynOff[pi]; k < partSynOff[pi + 1]; ++k) q.queryParts[pi].AddSynonymId(partSyn[k]);
			}
			q.SupressDuplicatesInSynonyms();
		}
		FtMergeStatuses::Statuses st;
		st.resize(f->totalDocs, false);
		if (excluded) {
			for (size_t i = 0; i < f->totalDocs; ++i) {
				if (excluded[i]) st.set(i);
			}
		}
		RdxContext ctx;
		ft::Merger<IdRelVec, ft::MergeData, uint32_t> merger(f->totalDocs, &f->cfg, st, f->nf, 0, /*inTransaction*/ true, ctx);
		Stats stats{f->words.data(), f->nf, f->avg.data(), f->hasRemoved ? f->removed.data() : nullptr};
		auto run = [&]() -> ft::MergeData {   // the selecter's dispatch on bm25Type (selecterimpl.h:615-624)
			switch (f->cfg.bm25Config.bm25Type) {
				case FTConfig::Bm25Config::Bm25Type::classic: return merger.Merge<Bm25Classic>(q, RankSortType(rankSortType), stats);
				case FTConfig::Bm25Config::Bm25Type::wordCount: return merger.Merge<TermCount>(q, RankSortType(rankSortType), stats);
				case FTConfig::Bm25Config::Bm25Type::rx: break;
			}
			return merger.Merge<Bm25Rx>(q, RankSortType(rankSortType), stats);
		};
		ft::MergeData md = run();
		for (size_t i = 0; i < md.size() && i < cap; ++i) {
			outId[i] = md[i].id.ToNumber();
			outProc[i] = md[i].proc;
			outField[i] = md[i].field;
			outNorm[i] = md[i].normalizedProc;
		}
		return long(md.size());
	} catch (const std::exception&) {
		return -1;
	}
}

long ref_ft_merge_phrases(void* h, size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost,
						  const uint8_t* needSum, const int* phraseNum, const int* distance, const uint32_t* subOff, const uint32_t* subWord,
						  const float* subProc, const uint8_t* excluded, int rankSortType, int32_t* outId, float* outProc, uint8_t* outField,
						  uint8_t* outNorm, size_t cap) {
	return ref_ft_merge_full(h, nTerms, 0, ops, boosts, termLenBoosts, fieldBoost, needSum, phraseNum, distance, subOff, subWord, subProc, 0, nullptr, nullptr,
							 nullptr, excluded, rankSortType, outId, outProc, outField, outNorm, cap);
}
long ref_ft_merge(void* h, size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost,
				  const uint8_t* needSum, const uint32_t* subOff, const uint32_t* subWord, const float* subProc, const uint8_t* excluded,
				  int rankSortType, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm, size_t cap) {
	return ref_ft_merge_phrases(h, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, nullptr, nullptr, subOff, subWord, subProc, excluded, rankSortType,
								outId, outProc, outField, outNorm, cap);
}

// The REAL packer: PackedIdRelVec::insert_back (idrelset.h:229-261) over IdRelType::pack / packWithoutArrayIdxs (idrelset.cc:8-68,
// 141-189).  Returns the byte count (bytes copied when they fit cap) and the position where array data starts.
size_t ref_ft_pack(size_t n, const uint32_t* doc, const uint32_t* posOff, const uint32_t* posField, const uint32_t* posPos,
				   const uint32_t* posArrayIdx, uint8_t* out, size_t cap, uint64_t* arrayFoundPos) {
	std::vector<IdRelType> v;
	v.reserve(n);
	for (size_t i = 0; i < n; ++i) {
		IdRelType rel(doc[i]);
		for (uint32_t j = posOff[i]; j < posOff[i + 1]; ++j) rel.Add(posPos[j], posField[j], posArrayIdx ? posArrayIdx[j] : 0);
		v.emplace_back(std::move(rel));
	}
	PackedIdRelVec packed;
	packed.insert_back(v.begin(), v.end());
	if (packed.data_.size() <= cap) std::copy(packed.data_.begin(), packed.data_.end(), out);
	*arrayFoundPos = uint64_t(packed.arrayFoundPos_);
	return packed.data_.size();
}

}  // extern "C"
