// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// The FT half of the drop-in boundary, EXECUTED: this TU is compiled inside a patched scratch copy of the reference's headers
// (integration/patches/0003-ft-fast-gpu-merger.patch applied, -DWITH_RXGPU -DRXGPU_IN_TREE) and drives, over the reference's OWN types,
//   CPU:  reindexer::ft::Merger<IdCont, ft::MergeData, uint32_t>::Merge<Bm25T>   (what Selector<IdCont>::mergeResults runs today)
//   GPU:  rxgpu::host::TryMergeOnGpu(...)                                          (what the patched mergeResults runs first)
// on the same ft::QueryMergeData<IdCont>, the same FTConfig / FtDslOpts / FtMergeStatuses::Statuses and a word table of
// PackedWordEntry<IdCont> exactly as DataHolder<IdCont>::words_ holds it (dataholder.h:28-60, 186-207), for IdCont = PackedIdRelVec
// (Optimization::Memory: the packed streams are decoded on the device) and IdRelVec (Optimization::CPU).
// tests/test_gpu_ft_seam.py compares the two ft::MergeData.  Output: oracle/_ref/libref_ft_seam.so (links librxgpu_host.so).
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "tools/float_comparison.h"
#include "core/ft/config/ftconfig.h"
#include "core/ft/ft_fast/selecterimpl.h"   // patched: includes rx_ft_seam.h

using namespace reindexer;

namespace {

// duck-typed like IndexText (indextext.h:245-258): NumWordsInField returns size_t there
struct Stats {
	const float* words;
	size_t nf;
	const float* avg;
	const uint8_t* removed;
	bool DocRemoved(uint32_t d) const noexcept { return removed[d] != 0; }
	size_t NumWordsInField(uint32_t d, uint32_t f) const noexcept { return size_t(words[d * nf + f]); }
	float AvgWordsCount(uint32_t f) const noexcept { return avg[f]; }
};

struct SeamRef {
	size_t nf, totalDocs = 0;
	std::vector<float> words, avg;
	std::vector<uint8_t> removed;
	std::vector<PackedWordEntry<PackedIdRelVec>> packedWords;
	std::vector<PackedWordEntry<IdRelVec>> plainWords;
	FTConfig cfg;
	std::shared_ptr<rxgpu::host::GpuFtMirror> packedMirror, plainMirror;
	std::string error;
	explicit SeamRef(size_t fields) : nf(fields), cfg(fields) {}
	Stats stats() const { return Stats{words.data(), nf, avg.data(), removed.data()}; }
};

template <typename IdCont>
const std::vector<PackedWordEntry<IdCont>>& wordsOf(const SeamRef& f) {
	if constexpr (std::is_same_v<IdCont, PackedIdRelVec>) {
		return f.packedWords;
	} else {
		return f.plainWords;
	}
}

// the query as Selector::Process hands it to mergeResults + the docsExcluded statuses, then fn(query, statuses)
template <typename IdCont, typename Fn>
long withQuery(SeamRef* f, size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost,
			   const uint8_t* needSum, const int* phraseNum, const int* distance, size_t nSynTerms, size_t nSyn, const uint32_t* synTermOff,
			   const uint32_t* partSynOff, const uint32_t* partSyn, const uint32_t* subOff, const uint32_t* subWord, const float* subProc, const uint8_t* excluded,
			   Fn&& fn) {
	const auto& words = wordsOf<IdCont>(*f);
	ft::QueryMergeData<IdCont> q;
	// query parts as Selector::Process puts them together (selecterimpl.h:482-572): terms with the same phraseNum >= 0 -> one PhraseResults
	int curPhraseNum = -1;
	ft::PhraseResults<IdCont> nextPhrase;
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
		ft::TermResults<IdCont> tr{FtDSLEntry(std::wstring(L"t") + std::to_wstring(t), o)};
		for (uint32_t s = subOff[t]; s < subOff[t + 1]; ++s) {
			WordIdType wid;
			wid.data = 0;
			wid.b.step_num = 1;   // a later commit step: only b.id addresses DataHolder::words_
			wid.SetID(int32_t(subWord[s]));
			tr.AddSubterm(words.at(subWord[s]).vids, std::string_view("w"), wid, subProc[s]);
		}
		q.totalORVids += tr.MaxVDocs();   // selecterimpl.h:546
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
	// multi-word synonyms (selecterimpl.h:427-466, 576-603), then what Selector::Process does in front of mergeResults (:606)
	for (size_t sy = 0; sy < nSyn; ++sy) {
		ft::Synonym<IdCont> syn;
		for (uint32_t k = synTermOff[sy]; k < synTermOff[sy + 1]; ++k) {
			const size_t t = nTerms + k;
			FtDslOpts o;
			o.op = OpType(ops[t]);
			o.boost = boosts[t];
			o.termLenBoost = termLenBoosts[t];
			o.fieldsOpts.resize(f->nf);
			for (size_t i = 0; i < f->nf; ++i) {
				o.fieldsOpts[i].boost = fieldBoost[t * f->nf + i];
				o.fieldsOpts[i].needSumRank = needSum[t * f->nf + i] != 0;
			}
			ft::TermResults<IdCont> tr{FtDSLEntry(std::wstring(L"s") + std::to_wstring(t), o)};
			for (uint32_t s = subOff[t]; s < subOff[t + 1]; ++s) {
				WordIdType wid;
				wid.data = 0;
				wid.b.step_num = 1;
				wid.SetID(int32_t(subWord[s]));
				tr.AddSubterm(words.at(subWord[s]).vids, std::string_view("w"), wid, subProc[s]);
			}
			q.totalORVids += tr.MaxVDocs();
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
	(void)nSynTerms;
	FtMergeStatuses::Statuses st;
	st.resize(f->totalDocs, false);
	if (excluded) {
		for (size_t i = 0; i < f->totalDocs; ++i) {
			if (excluded[i]) st.set(i);
		}
	}
	return fn(q, st);
}

template <typename IdCont>
long mergeImpl(SeamRef* f, bool gpu, size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost,
			   const uint8_t* needSum, const int* phraseNum, const int* distance, size_t nSynTerms, size_t nSyn, const uint32_t* synTermOff,
			   const uint32_t* partSynOff, const uint32_t* partSyn, const uint32_t* subOff, const uint32_t* subWord, const float* subProc, const uint8_t* excluded,
			   int rankSortType, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm, size_t cap) {
	return withQuery<IdCont>(f, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, phraseNum, distance, nSynTerms, nSyn, synTermOff, partSynOff, partSyn, subOff,
							 subWord, subProc, excluded, [&](ft::QueryMergeData<IdCont>& q, FtMergeStatuses::Statuses& st) -> long {
	bool declined = false;
	auto run = [&]() -> ft::MergeData {
		ft::MergeData out;
		if (gpu) {
			const auto& mirror = std::is_same_v<IdCont, PackedIdRelVec> ? f->packedMirror : f->plainMirror;
			declined = !rxgpu::host::TryMergeOnGpu(mirror.get(), f->cfg, f->totalDocs, q, RankSortType(rankSortType), st, /*inTransaction*/ false, out);
			return out;
		}
		RdxContext ctx;
		ft::Merger<IdCont, ft::MergeData, uint32_t> merger(f->totalDocs, &f->cfg, st, f->nf, 0, /*inTransaction*/ true, ctx);
		const Stats stats = f->stats();
		switch (f->cfg.bm25Config.bm25Type) {   // selecterimpl.h:615-624
			case FTConfig::Bm25Config::Bm25Type::classic: return merger.template Merge<Bm25Classic>(q, RankSortType(rankSortType), stats);
			case FTConfig::Bm25Config::Bm25Type::wordCount: return merger.template Merge<TermCount>(q, RankSortType(rankSortType), stats);
			case FTConfig::Bm25Config::Bm25Type::rx: break;
		}
		return merger.template Merge<Bm25Rx>(q, RankSortType(rankSortType), stats);
	};
	ft::MergeData md = run();
	if (declined) return -2;
	for (size_t i = 0; i < md.size() && i < cap; ++i) {
		outId[i] = md[i].id.ToNumber();
		outProc[i] = md[i].proc;
		outField[i] = md[i].field;
		outNorm[i] = md[i].normalizedProc;
	}
	return long(md.size());
	});
}

// The merge behind highlight() / snippet(): MergedDataType = MergeDataAreas<Area> (selecterimpl.h:611-628 with that instantiation).  gpu: the
// adapter the patched mergeResults calls first; else the reference's ft::Merger.  Per merged document i and field fld the areas come back twice:
// raw (AreasInField::data_ as the merge left it, GetAreasRaw) and committed (GetAreas: sorted by start, neighbours joined — what the engine
// reads): areaOff [2][n * nf + 1] into areas [..][3] = start, end, arrayIdx.
template <typename IdCont>
long mergeAreasImpl(SeamRef* f, bool gpu, int maxAreasInDoc, size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost,
					const uint8_t* needSum, const int* phraseNum, const int* distance, const uint32_t* subOff, const uint32_t* subWord, const float* subProc,
					const uint8_t* excluded, int rankSortType, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm, size_t cap,
					uint32_t* rawOff, uint32_t* rawAreas, uint32_t* comOff, uint32_t* comAreas, size_t areaCap) {
	return withQuery<IdCont>(f, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, phraseNum, distance, 0, 0, nullptr, nullptr, nullptr, subOff, subWord, subProc,
							 excluded, [&](ft::QueryMergeData<IdCont>& q, FtMergeStatuses::Statuses& st) -> long {
		using MD = ft::MergeDataAreas<Area>;
		bool declined = false;
		auto run = [&]() -> MD {
			if (gpu) {
				MD out;
				const auto& mirror = std::is_same_v<IdCont, PackedIdRelVec> ? f->packedMirror : f->plainMirror;
				declined = !rxgpu::host::TryMergeOnGpu(mirror.get(), f->cfg, f->totalDocs, q, RankSortType(rankSortType), st, /*inTransaction*/ false, out, maxAreasInDoc);
				return out;
			}
			RdxContext ctx;
			ft::Merger<IdCont, MD, uint32_t> merger(f->totalDocs, &f->cfg, st, f->nf, maxAreasInDoc, /*inTransaction*/ true, ctx);
			const Stats stats = f->stats();
			switch (f->cfg.bm25Config.bm25Type) {
				case FTConfig::Bm25Config::Bm25Type::classic: return merger.template Merge<Bm25Classic>(q, RankSortType(rankSortType), stats);
				case FTConfig::Bm25Config::Bm25Type::wordCount: return merger.template Merge<TermCount>(q, RankSortType(rankSortType), stats);
				case FTConfig::Bm25Config::Bm25Type::rx: break;
			}
			return merger.template Merge<Bm25Rx>(q, RankSortType(rankSortType), stats);
		};
		MD md = run();
		if (declined) return -2;
		size_t nRaw = 0, nCom = 0;
		rawOff[0] = comOff[0] = 0;
		for (size_t i = 0; i < md.size() && i < cap; ++i) {
			outId[i] = md[i].id.ToNumber();
			outProc[i] = md[i].proc;
			outField[i] = md[i].field;
			outNorm[i] = md[i].normalizedProc;
			auto& doc = md.vectorAreas.at(md[i].areaIndex);
			for (size_t fld = 0; fld < f->nf; ++fld) {
				if (const auto* raw = doc.GetAreasRaw(unsigned(fld))) {
					for (const Area& a : raw->GetData()) {
						if (nRaw >= areaCap) throw std::runtime_error("area buffer too small");
						rawAreas[nRaw * 3] = a.start;
						rawAreas[nRaw * 3 + 1] = a.end;
						rawAreas[nRaw * 3 + 2] = a.arrayIdx;
						++nRaw;
					}
				}
				rawOff[i * f->nf + fld + 1] = uint32_t(nRaw);
			}
			for (size_t fld = 0; fld < f->nf; ++fld) {
				if (const auto* com = doc.GetAreas(unsigned(fld))) {   // commits
					for (const Area& a : com->GetData()) {
						if (nCom >= areaCap) throw std::runtime_error("area buffer too small");
						comAreas[nCom * 3] = a.start;
						comAreas[nCom * 3 + 1] = a.end;
						comAreas[nCom * 3 + 2] = a.arrayIdx;
						++nCom;
					}
				}
				comOff[i * f->nf + fld + 1] = uint32_t(nCom);
			}
		}
		return long(md.size());
	});
}

}  // namespace

extern "C" {

void* ref_seam_create(size_t numFields) { return new SeamRef(numFields); }
void ref_seam_destroy(void* h) { delete static_cast<SeamRef*>(h); }
const char* ref_seam_last_error(void* h) { return static_cast<SeamRef*>(h)->error.c_str(); }

void ref_seam_set_docs(void* h, size_t totalDocs, const float* words, const float* avg, const uint8_t* removed) {
	auto* f = static_cast<SeamRef*>(h);
	f->totalDocs = totalDocs;
	f->words.assign(words, words + totalDocs * f->nf);
	f->avg.assign(avg, avg + f->nf);
	f->removed.assign(totalDocs, 0);
	if (removed) f->removed.assign(removed, removed + totalDocs);
}

// one dictionary word, appended the way DataHolder::Process fills words_[id].vids (insert_back for the packed form)
void ref_seam_set_word(void* h, uint32_t wordId, size_t n, const uint32_t* doc, const uint32_t* posOff, const uint32_t* posField,
					   const uint32_t* posPos, const uint32_t* posArrayIdx) {
	auto* f = static_cast<SeamRef*>(h);
	if (f->packedWords.size() <= wordId) {
		f->packedWords.resize(wordId + 1);
		f->plainWords.resize(wordId + 1);
	}
	IdRelVec vec;
	vec.reserve(n);
	for (size_t i = 0; i < n; ++i) {
		IdRelType rel(doc[i]);
		for (uint32_t j = posOff[i]; j < posOff[i + 1]; ++j) rel.Add(posPos[j], posField[j], posArrayIdx ? posArrayIdx[j] : 0);
		vec.emplace_back(std::move(rel));
	}
	f->packedWords[wordId].vids.clear();
	f->packedWords[wordId].vids.insert_back(vec.begin(), vec.end());
	f->plainWords[wordId].vids = std::move(vec);
}

// same layout as ref_ft_set_config / ref_ft_set_bm25_type (ref_ft_shim.cc)
void ref_seam_set_config(void* h, const double* cfgD, const int* cfgI, const double* fieldCfg, int bm25Type) {
	auto* f = static_cast<SeamRef*>(h);
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
	using T = FTConfig::Bm25Config::Bm25Type;
	f->cfg.bm25Config.bm25Type = bm25Type == 1 ? T::classic : bm25Type == 2 ? T::wordCount : T::rx;
}

// what the patched IndexText::commitFulltextImpl does at the end of a commit: mirror (re)created, statistics and changed words handed over.
// Returns the number of words on the device, -1 on error (message in ref_seam_last_error).
// n_devices > 1: the mirror over a device list (what RX_GPU_FT_INDEXES=0-7 makes SyncGpuFtMirror construct: document-range shards).
long ref_seam_commit_devices(void* h, const int* devices, size_t n_devices) {
	auto* f = static_cast<SeamRef*>(h);
	try {
		const Stats stats = f->stats();
		for (int packed = 0; packed < 2; ++packed) {
			auto& mirror = packed ? f->packedMirror : f->plainMirror;
			if (!mirror) {
				if (n_devices > 1) {
					mirror = std::make_shared<rxgpu::host::GpuFtMirror>(f->nf, std::vector<int>(devices, devices + n_devices));
				} else {
					mirror = std::make_shared<rxgpu::host::GpuFtMirror>(f->nf, n_devices ? devices[0] : 0);
				}
			}
			mirror->SyncDocs(f->totalDocs, stats);
			if (packed) {
				mirror->SyncWords(f->packedWords);
			} else {
				mirror->SyncWords(f->plainWords);
			}
		}
		return long(f->packedMirror->SyncedWords());
	} catch (const std::exception& e) {
		f->error = e.what();
		return -1;
	}
}
long ref_seam_commit(void* h, int device) { return ref_seam_commit_devices(h, &device, 1); }

// packed: 1 = QueryMergeData<PackedIdRelVec>, 0 = <IdRelVec>; gpu: 1 = TryMergeOnGpu, 0 = the reference's ft::Merger.
// Returns the result count, -1 on an exception, -2 when the GPU branch declined the query (the CPU merger would run).
// phraseNum / distance: FtDslOpts::phraseNum (-1: a plain term) / FtDslOpts::distance per term, or null; synonyms as in ref_ft_merge_full
// (ref_ft_shim.cc): per-term arrays hold nTerms + nSynTerms entries.
long ref_seam_merge_full(void* h, int packed, int gpu, size_t nTerms, size_t nSynTerms, const int* ops, const float* boosts, const float* termLenBoosts,
						 const float* fieldBoost, const uint8_t* needSum, const int* phraseNum, const int* distance, const uint32_t* subOff,
						 const uint32_t* subWord, const float* subProc, size_t nSyn, const uint32_t* synTermOff, const uint32_t* partSynOff,
						 const uint32_t* partSyn, const uint8_t* excluded, int rankSortType, int32_t* outId, float* outProc, uint8_t* outField,
						 uint8_t* outNorm, size_t cap) {
	auto* f = static_cast<SeamRef*>(h);
	try {
		if (packed) {
			return mergeImpl<PackedIdRelVec>(f, gpu != 0, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, phraseNum, distance, nSynTerms, nSyn, synTermOff,
											 partSynOff, partSyn, subOff, subWord, subProc, excluded, rankSortType, outId, outProc, outField, outNorm, cap);
		}
		return mergeImpl<IdRelVec>(f, gpu != 0, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, phraseNum, distance, nSynTerms, nSyn, synTermOff, partSynOff,
								   partSyn, subOff, subWord, subProc, excluded, rankSortType, outId, outProc, outField, outNorm, cap);
	} catch (const std::exception& e) {
		f->error = e.what();
		return -1;
	}
}
long ref_seam_merge_areas(void* h, int packed, int gpu, int maxAreasInDoc, size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts,
						  const float* fieldBoost, const uint8_t* needSum, const int* phraseNum, const int* distance, const uint32_t* subOff, const uint32_t* subWord,
						  const float* subProc, const uint8_t* excluded, int rankSortType, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm,
						  size_t cap, uint32_t* rawOff, uint32_t* rawAreas, uint32_t* comOff, uint32_t* comAreas, size_t areaCap) {
	auto* f = static_cast<SeamRef*>(h);
	try {
		if (packed) {
			return mergeAreasImpl<PackedIdRelVec>(f, gpu != 0, maxAreasInDoc, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, phraseNum, distance, subOff, subWord,
												  subProc, excluded, rankSortType, outId, outProc, outField, outNorm, cap, rawOff, rawAreas, comOff, comAreas, areaCap);
		}
		return mergeAreasImpl<IdRelVec>(f, gpu != 0, maxAreasInDoc, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, phraseNum, distance, subOff, subWord, subProc,
										excluded, rankSortType, outId, outProc, outField, outNorm, cap, rawOff, rawAreas, comOff, comAreas, areaCap);
	} catch (const std::exception& e) {
		f->error = e.what();
		return -1;
	}
}
long ref_seam_merge_phrases(void* h, int packed, int gpu, size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts,
							const float* fieldBoost, const uint8_t* needSum, const int* phraseNum, const int* distance, const uint32_t* subOff,
							const uint32_t* subWord, const float* subProc, const uint8_t* excluded, int rankSortType, int32_t* outId, float* outProc,
							uint8_t* outField, uint8_t* outNorm, size_t cap) {
	return ref_seam_merge_full(h, packed, gpu, nTerms, 0, ops, boosts, termLenBoosts, fieldBoost, needSum, phraseNum, distance, subOff, subWord, subProc, 0, nullptr,
							   nullptr, nullptr, excluded, rankSortType, outId, outProc, outField, outNorm, cap);
}
long ref_seam_merge(void* h, int packed, int gpu, size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts,
					const float* fieldBoost, const uint8_t* needSum, const uint32_t* subOff, const uint32_t* subWord, const float* subProc,
					const uint8_t* excluded, int rankSortType, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm, size_t cap) {
	return ref_seam_merge_phrases(h, packed, gpu, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, nullptr, nullptr, subOff, subWord, subProc,
								  excluded, rankSortType, outId, outProc, outField, outNorm, cap);
}

}  // extern "C"
