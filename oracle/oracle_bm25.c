/* TEST INFRASTRUCTURE ONLY — the CHECKER for the GPU BM25 merge, never the product.
 *
 * Plain-C restatement of the ft_fast single-term merge of Restream/reindexer v5.15.0:
 *   Bm25Rx::IDF / Get                 cpp_src/core/ft/bm25.h:8-36
 *   FTFieldConfig::pos2rank / bound   cpp_src/core/ft/config/ftconfig.h:127-148
 *   calcTermRankImpl                  cpp_src/core/ft/ft_fast/phrasemergerimpl.h:13-81
 *   Merger::mergeSimple               cpp_src/core/ft/ft_fast/mergerimpl.h:194-250   (mergeLimit, max over sub-terms, first max wins)
 *   addFullMatchBoost                 cpp_src/core/ft/ft_fast/merger.h:100-109
 *   postProcessResults                cpp_src/core/ft/ft_fast/merger.h:111-155       (minRank swap-remove, uint8 normalisation)
 * Every float/double conversion follows the reference's declared types (double Bm25Rx, float bound() arguments, float ranks).
 *
 * PARITY PINNED twice: (1) by the reference's own golden vectors — the `debug_rank()` strings of
 * cpp_src/gtests/tests/unit/ft/ft_generic.cc:326-443 (bm25_norm, position_rank, term_len_boost, term_rank to 7-8 digits) are
 * replayed in tests/test_bm25_oracle.py; (2) against the REAL reindexer::ft::Merger, compiled in place from the reference tree into
 * oracle/_ref/libref_ft.so (oracle/ref/ref_ft_shim.cc, `make -C oracle ref`): same documents in the same merge order, same raw-rank
 * bits, fields and uint8 ranks, for single-term and multi-term queries (tests/test_bm25_oracle.py).
 *
 * Posting layout (what IndexText hands to the merger, flattened): a sub-term's posting list is
 *   doc[u32]  ent_off[u32, n+1]  and per (doc, field) entry:  ent_field[u8]  ent_tf[u32]  ent_first_pos[u32]
 * with entries of one doc sorted by field — the information calcTermRank extracts from IdRelType::Pos().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	double k1, b;                 /* FTConfig::Bm25Config (ftconfig.h:199-203): 2.0, 0.75 */
	double summation_ratio;       /* summationRanksByFieldsRatio (ftconfig.h:210): 0.0 */
	double full_match_boost;      /* 1.1 */
	int min_rank;                 /* 5 */
	uint32_t merge_limit;         /* 20000 */
	uint32_t num_fields;
	/* per field: FTFieldConfig (ftconfig.h:118-124) */
	const double* bm25_boost;
	const double* bm25_weight;
	const double* term_len_boost;
	const double* term_len_weight;
	const double* position_boost;
	const double* position_weight;
	int bm25_type;                /* FTConfig::Bm25Config::bm25Type (ftconfig.h:200-203): 0 = rx (default), 1 = classic, 2 = wordCount */
} orc_ft_config;

typedef struct {
	float boost;                  /* FtDslOpts::boost */
	float term_len_boost;         /* FtDslOpts::termLenBoost */
	const float* field_boost;     /* FtDslFieldOpts::boost per field */
	const uint8_t* need_sum_rank; /* FtDslFieldOpts::needSumRank per field */
} orc_ft_term_opts;

typedef struct {
	uint64_t n;
	const uint32_t* doc;
	const uint32_t* ent_off;
	const uint8_t* ent_field;
	const uint32_t* ent_tf;
	const uint32_t* ent_first_pos;
	float proc;                   /* SubtermResults::Proc() */
} orc_ft_postings;

/* bm25.h:19-26 */
double orc_bm25rx_idf(double total_docs, double matched_docs) {
	double f = log((total_docs - matched_docs + 1) / matched_docs) / log(1 + total_docs);
	if (f < 0.2) f = 0.2;
	return f;
}
/* bm25.h:13-16 (TF = termCountInDoc) */
double orc_bm25rx_get(double idf, double k1, double b, double term_count, double words_in_doc, double avg_doc_len) {
	const double tf = term_count;
	return idf * tf * (k1 + 1.0) / (tf + k1 * (1.0 - b + b * words_in_doc / avg_doc_len));
}
/* The three calculators behind Bm25Calculator<BM> (bm25.h:8-68), selected like the selecter does from bm25Config.bm25Type:
 * rx: IDF saturated at 0.2, TF = count; classic: IDF = ln(N / (M + 1)) + 1, TF = count / wordsInDoc; wordCount: the count itself, IDF 0 */
double orc_bm25_idf(int type, double total_docs, double matched_docs) {
	if (type == 1) return log(total_docs / (matched_docs + 1)) + 1;
	if (type == 2) return 0.0;
	return orc_bm25rx_idf(total_docs, matched_docs);
}
double orc_bm25_get(int type, double idf, double k1, double b, double term_count, double words_in_doc, double avg_doc_len) {
	if (type == 2) return term_count;
	const double tf = type == 1 ? term_count / words_in_doc : term_count;
	return idf * tf * (k1 + 1.0) / (tf + k1 * (1.0 - b + b * words_in_doc / avg_doc_len));
}
/* ftconfig.h:127-144 */
float orc_pos2rank(unsigned pos) {
	if (pos <= 10) return (float)(1.0 - (pos / 100.0));
	if (pos <= 100) return (float)(0.9 - (pos / 1000.0));
	if (pos <= 1000) return (float)(0.8 - (pos / 10000.0));
	if (pos <= 10000) return (float)(0.7 - (pos / 100000.0));
	if (pos <= 100000) return (float)(0.6 - (pos / 1000000.0));
	return 0.5f;
}
/* ftconfig.h:146: float bound(float k, float weight, float boost) { return (1.0 - weight) + k * boost * weight; } */
float orc_bound(float k, float weight, float boost) { return (float)((1.0 - (double)weight) + (double)(k * boost * weight)); }

static int cmp_desc(const void* a, const void* b) {
	const float x = *(const float*)a, y = *(const float*)b;
	return (x < y) - (x > y);
}

/* calcTermRankImpl (phrasemergerimpl.h:13-81) for one posting.  Returns the rank, *field = field with the max rank.
 * Optional outputs of the winning field for the golden strings: bm25_norm, term_len_boost, position_rank. */
float orc_calc_term_rank(const orc_ft_config* cfg, const orc_ft_term_opts* opts, double idf, float proc, uint32_t nent,
						 const uint8_t* ent_field, const uint32_t* ent_tf, const uint32_t* ent_first_pos, const float* words_in_field /* of this doc, [num_fields] */,
						 const float* avg_words, uint8_t* field, float* out_bm25_norm, float* out_term_len_boost, float* out_position_rank) {
	uint8_t field_with_max = 0;
	float ranks[64];
	size_t nranks = 0;
	int need_sum_winner = 0;
	float term_rank = 0.f, bm25_norm = 0.f, tlb = 0.f, prank = 0.f;
	const int need_sum = cfg->summation_ratio > 0.0;
	for (uint32_t e = 0; e < nent; ++e) {
		const unsigned f = ent_field[e];
		if (opts->field_boost[f] == 0.0f) continue;
		const float bm25 = (float)orc_bm25_get(cfg->bm25_type, idf, cfg->k1, cfg->b, (double)ent_tf[e], (double)words_in_field[f], (double)avg_words[f]);
		const float norm_bm25 = orc_bound(bm25, (float)cfg->bm25_weight[f], (float)cfg->bm25_boost[f]);
		prank = orc_bound(orc_pos2rank(ent_first_pos[e]), (float)cfg->position_weight[f], (float)cfg->position_boost[f]);
		tlb = orc_bound(opts->term_len_boost, (float)cfg->term_len_weight[f], (float)cfg->term_len_boost[f]);
		const float tmp = opts->field_boost[f] * norm_bm25 * tlb * prank;
		if (tmp > term_rank) {
			field_with_max = (uint8_t)f;
			term_rank = tmp;
			bm25_norm = norm_bm25;
			need_sum_winner = opts->need_sum_rank[f];
			if (out_term_len_boost) *out_term_len_boost = tlb;
			if (out_position_rank) *out_position_rank = prank;
		}
		if (opts->need_sum_rank[f] && nranks < 64) ranks[nranks++] = tmp;
	}
	if (term_rank > 0.0f && need_sum) {
		qsort(ranks, nranks, sizeof(float), cmp_desc);
		float k = (float)cfg->summation_ratio;
		for (size_t i = need_sum_winner ? 1 : 0; i < nranks; ++i) {
			term_rank += (k * ranks[i]);
			k = (float)((double)k * cfg->summation_ratio);
		}
	}
	term_rank = opts->boost * proc * term_rank;
	*field = field_with_max;
	if (out_bm25_norm) *out_bm25_norm = bm25_norm;
	return term_rank;
}

/* postProcessResults (merger.h:111-155) on the merged rows: minRank swap-remove on the RAW rank, uint8 normalisation, optional rank sort */
size_t orc_ft_post_process(const orc_ft_config* cfg, size_t n, int sort_by_rank, uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint8_t* out_norm) {
	/* postProcessResults */
	float max_proc = 0.0f;
	for (size_t i = 0; i < n; ++i) max_proc = out_proc[i] > max_proc ? out_proc[i] : max_proc;
	const float scaling = (float)(max_proc > 255 ? 255.0 / (double)max_proc : 1.0);
	const float min_proc = (float)cfg->min_rank;
	size_t passed = n;
	while (passed > 0 && out_proc[passed - 1] < min_proc) passed--;
	for (size_t i = 0; i + 1 < passed; i++) {
		if (out_proc[i] < min_proc) {
			out_doc[i] = out_doc[passed - 1];
			out_proc[i] = out_proc[passed - 1];
			out_field[i] = out_field[passed - 1];
			passed--;
			while (passed > i && out_proc[passed - 1] < min_proc) passed--;
		}
	}
	n = passed;
	for (size_t i = 0; i < n; ++i) {
		out_norm[i] = (uint8_t)(out_proc[i] * scaling);
		out_proc[i] = (float)out_norm[i];
	}
	if (sort_by_rank) {   /* pdqsort is unstable in the reference; only (set, uint8 rank) is contractual — we sort stably */
		for (size_t i = 1; i < n; ++i) {   /* insertion sort is fine for the test sizes; large cases use the rank-only comparison */
			const uint32_t d = out_doc[i];
			const float pr = out_proc[i];
			const uint8_t f = out_field[i], nm = out_norm[i];
			size_t j = i;
			while (j > 0 && out_norm[j - 1] < nm) {
				out_doc[j] = out_doc[j - 1];
				out_proc[j] = out_proc[j - 1];
				out_field[j] = out_field[j - 1];
				out_norm[j] = out_norm[j - 1];
				--j;
			}
			out_doc[j] = d;
			out_proc[j] = pr;
			out_field[j] = f;
			out_norm[j] = nm;
		}
	}
	return n;
}

/* Merger::Merge for a Simple() query (one OR term with nsub sub-terms, already sorted by proc desc — SortSubterms):
 * mergeSimple + addFullMatchBoost(1) + postProcessResults.  total_docs = vdocs incl. the empty sentinel doc 0.
 * words: [total_docs][num_fields] float (VDoc::wordCounts_), removed: [total_docs] (or NULL), excluded: [total_docs] (or NULL).
 * sort_by_rank: RankOnly / IDAndPositions (sorted by normalizedProc desc; ties in merge order) vs RankAndID / IDOnly (merge order).
 * Returns the number of results; out_* sized >= min(merge_limit, total postings). */
size_t orc_ft_merge_simple(const orc_ft_config* cfg, const orc_ft_term_opts* opts, uint64_t total_docs, const float* words, const float* avg_words,
						   const uint8_t* removed, const uint8_t* excluded, const orc_ft_postings* subs, uint32_t nsub, int sort_by_rank,
						   uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint8_t* out_norm) {
	uint64_t total_or_vids = 0;
	for (uint32_t s = 0; s < nsub; ++s) total_or_vids += subs[s].n;
	if (nsub == 0 || total_docs == 0) return 0;
	uint64_t max_merged = cfg->merge_limit < total_or_vids ? cfg->merge_limit : total_or_vids;   /* Merge(): min(mergeLimit, totalORVids); init(): min(.., mergeLimit) */
	const int trivial = nsub == 1;   /* Trivial(): idoffsets_ is not allocated => docAdded() is always false */
	uint32_t* idoff = NULL;
	if (!trivial) {
		idoff = (uint32_t*)malloc(total_docs * sizeof(uint32_t));
		for (uint64_t i = 0; i < total_docs; ++i) idoff[i] = (uint32_t)max_merged;
	}
	size_t n = 0;
	for (uint32_t s = 0; s < nsub; ++s) {
		const orc_ft_postings* p = &subs[s];
		const double idf = orc_bm25_idf(cfg->bm25_type, (double)(total_docs - 1), (double)p->n);   /* "first doc is always empty" */
		for (uint64_t i = 0; i < p->n; ++i) {
			const uint32_t d = p->doc[i];
			if ((excluded && excluded[d]) || (removed && removed[d])) continue;
			const int added = !trivial && idoff[d] != (uint32_t)max_merged;
			if (!added && n >= max_merged) continue;
			uint8_t field;
			const uint32_t e0 = p->ent_off[i], e1 = p->ent_off[i + 1];
			const float rank = orc_calc_term_rank(cfg, opts, idf, p->proc, e1 - e0, p->ent_field + e0, p->ent_tf + e0, p->ent_first_pos + e0,
												  words + (size_t)d * cfg->num_fields, avg_words, &field, NULL, NULL, NULL);
			if (rank == 0.0f) continue;
			if (!added) {
				out_doc[n] = d;
				out_proc[n] = rank;
				out_field[n] = field;
				if (idoff) idoff[d] = (uint32_t)n;
				++n;
			} else {
				const uint32_t o = idoff[d];
				if (out_proc[o] < rank) {
					out_proc[o] = rank;
					out_field[o] = field;
				}
			}
		}
	}
	free(idoff);
	/* addFullMatchBoost(1) */
	for (size_t i = 0; i < n; ++i) {
		if (words[(size_t)out_doc[i] * cfg->num_fields + out_field[i]] == (float)1) out_proc[i] = (float)((double)out_proc[i] * cfg->full_match_boost);
	}
	return orc_ft_post_process(cfg, n, sort_by_rank, out_doc, out_proc, out_field, out_norm);
}

/* ================================================================================================================
 * Multi-term queries: Merger::Merge for queries that are not Simple() (mergerimpl.h:466-566), terms only
 * (no phrases, no multi-word synonyms — those are outside SURVEY §8):
 *   buildRestrictingBitmask   mergerimpl.h:326-384   (~docsExcluded, AND-term masks, NOT-term exclusion)
 *   calcTermBitmask           mergerimpl.h:252-274
 *   estimateNumDocsInMerge    merger.h:239-267
 *   preselectMostRelevantDocs mergerimpl.h:386-464   + calcTermScores :289-324   (uint16 saturating pre-score, counting sort)
 *   mergeTerm                 mergerimpl.h:107-192   + PositionsDistance :20-37, switchToNextWord merger.h:218-226
 *   canBeBoostedByFullMatch   mergerimpl.h:527-531,  addFullMatchBoost / postProcessResults merger.h:100-155
 * PARITY PINNED against the real ft::Merger compiled in place (oracle/_ref/libref_ft.so, tests/test_bm25_oracle.py).
 *
 * Postings carry the positions here: per posting a run [pos_off[i], pos_off[i+1]) of 64-bit PosType words
 * (idrelset.h:14-32: pos | arrayIdx << 28 | field << 56), ascending — IdRelType::Pos() verbatim.
 */
typedef struct {
	uint64_t n;
	const uint32_t* doc;
	const uint32_t* pos_off;
	const uint64_t* fpos;
	float proc;
} orc_ft_ppostings;

typedef struct {
	int op; /* OpType: 1 OR, 2 AND, 3 NOT (core/type_consts.h) */
	orc_ft_term_opts opts;
	uint32_t nsub;
	const orc_ft_ppostings* subs; /* in SortSubterms order (proc desc, stable) */
} orc_ft_term;

static inline uint32_t pt_field(uint64_t p) { return (uint32_t)(p >> 56); }
static inline uint32_t pt_pos(uint64_t p) { return (uint32_t)(p & ((1u << 28) - 1)); }
static inline uint32_t pt_full_pos(uint64_t p) { return (uint32_t)p; }            /* uint32_t fullPos() { return fpos_; } — truncates */
static inline uint32_t pt_full_field(uint64_t p) { return (uint32_t)(p >> 28); }  /* uint32_t fullField() { return fpos_ >> posBits; } */

/* mergerimpl.h:20-37 */
unsigned orc_positions_distance(const uint64_t* a, uint32_t na, const uint64_t* b, uint32_t nb) {
	unsigned res = 0xFFFFFFFFu;
	uint32_t i = 0, j = 0;
	while (i < na && j < nb) {
		const int sign = pt_full_pos(a[i]) > pt_full_pos(b[j]);
		if (pt_full_field(a[i]) == pt_full_field(b[j])) {
			const unsigned dst = sign ? pt_full_pos(a[i]) - pt_full_pos(b[j]) : pt_full_pos(b[j]) - pt_full_pos(a[i]);
			if (dst < res) {
				res = dst;
				if (res <= 1) break;
			}
		}
		if (sign) ++j; else ++i;
	}
	return res == 0xFFFFFFFFu ? 0 : res;
}

/* calcTermRank over IdRelType::Pos(): group by field -> (tf, first pos) -> orc_calc_term_rank */
static float rank_from_positions(const orc_ft_config* cfg, const orc_ft_term_opts* opts, double idf, float proc, const uint64_t* pos, uint32_t npos,
								 const float* words_in_field, const float* avg_words, uint8_t* field) {
	uint8_t ef[64];
	uint32_t tf[64], fp[64];
	uint32_t ne = 0;
	for (uint32_t i = 0; i < npos;) {
		const uint32_t f = pt_field(pos[i]);
		uint32_t j = i + 1;
		while (j < npos && pt_field(pos[j]) == f) ++j;
		if (ne < 64) {
			ef[ne] = (uint8_t)f;
			tf[ne] = j - i;
			fp[ne] = pt_pos(pos[i]);
			++ne;
		}
		i = j;
	}
	return orc_calc_term_rank(cfg, opts, idf, proc, ne, ef, tf, fp, words_in_field, avg_words, field, NULL, NULL, NULL);
}

typedef struct {
	const uint64_t* last; uint32_t nlast;   /* lastTermPositions */
	const uint64_t* next; uint32_t nnext;   /* nextTermPositions */
	float rank;
	uint16_t last_term_counted, terms_counter;
} orc_doc_ext;

static inline int bit_get(const uint8_t* m, uint64_t i) { return m[i]; }

size_t orc_ft_merge_query(const orc_ft_config* cfg, double distance_boost, double distance_weight, const orc_ft_term* terms, uint32_t nterms,
						  uint64_t total_docs, const float* words, const float* avg_words, const uint8_t* removed, const uint8_t* excluded,
						  int sort_by_rank, uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint8_t* out_norm, uint8_t* out_preselected) {
	if (out_preselected) *out_preselected = 0;
	/* Empty() */
	if (nterms == 0 || (nterms == 1 && terms[0].op == 3) || total_docs == 0) return 0;
	const uint32_t nf = cfg->num_fields;
	uint64_t total_vids = 0;   /* totalORVids: MaxVDocs of EVERY term (selecterimpl.h:546) */
	for (uint32_t t = 0; t < nterms; ++t)
		for (uint32_t s = 0; s < terms[t].nsub; ++s) total_vids += terms[t].subs[s].n;
	const uint64_t max_merged = cfg->merge_limit < total_vids ? cfg->merge_limit : total_vids;

	if (nterms == 1) {   /* Simple(): flatten positions to (field, tf, first pos) entries and run mergeSimple */
		const orc_ft_term* T = &terms[0];
		orc_ft_postings* flat = (orc_ft_postings*)calloc(T->nsub ? T->nsub : 1, sizeof(orc_ft_postings));
		void** owned = (void**)calloc((size_t)T->nsub * 4 + 1, sizeof(void*));
		for (uint32_t s = 0; s < T->nsub; ++s) {
			const orc_ft_ppostings* p = &T->subs[s];
			const uint64_t npos = p->n ? p->pos_off[p->n] : 0;
			uint32_t* eo = (uint32_t*)malloc((p->n + 1) * sizeof(uint32_t));
			uint8_t* ef = (uint8_t*)malloc(npos + 1);
			uint32_t* et = (uint32_t*)malloc((npos + 1) * sizeof(uint32_t));
			uint32_t* ep = (uint32_t*)malloc((npos + 1) * sizeof(uint32_t));
			uint32_t ne = 0;
			for (uint64_t i = 0; i < p->n; ++i) {
				eo[i] = ne;
				for (uint32_t a = p->pos_off[i]; a < p->pos_off[i + 1];) {
					const uint32_t f = pt_field(p->fpos[a]);
					uint32_t b = a + 1;
					while (b < p->pos_off[i + 1] && pt_field(p->fpos[b]) == f) ++b;
					ef[ne] = (uint8_t)f;
					et[ne] = b - a;
					ep[ne] = pt_pos(p->fpos[a]);
					++ne;
					a = b;
				}
			}
			eo[p->n] = ne;
			flat[s].n = p->n; flat[s].doc = p->doc; flat[s].ent_off = eo; flat[s].ent_field = ef; flat[s].ent_tf = et; flat[s].ent_first_pos = ep;
			flat[s].proc = p->proc;
			owned[s * 4] = eo; owned[s * 4 + 1] = ef; owned[s * 4 + 2] = et; owned[s * 4 + 3] = ep;
		}
		const size_t n = orc_ft_merge_simple(cfg, &T->opts, total_docs, words, avg_words, removed, excluded, flat, T->nsub, sort_by_rank, out_doc, out_proc,
											 out_field, out_norm);
		for (uint32_t s = 0; s < T->nsub * 4; ++s) free(owned[s]);
		free(owned);
		free(flat);
		return n;
	}

	/* ---- buildRestrictingBitmask ---- */
	uint8_t* mask = (uint8_t*)malloc(total_docs);
	uint8_t* tmask = (uint8_t*)malloc(total_docs);
	for (uint64_t i = 0; i < total_docs; ++i) mask[i] = !(excluded && excluded[i]);
	for (uint32_t t = 0; t < nterms; ++t) {
		if (terms[t].op != 2) continue;
		memset(tmask, 0, total_docs);
		int all_pos = 1;
		for (uint32_t f = 0; f < nf; ++f) all_pos &= terms[t].opts.field_boost[f] != 0.0f;
		for (uint32_t s = 0; s < terms[t].nsub; ++s) {
			const orc_ft_ppostings* p = &terms[t].subs[s];
			for (uint64_t i = 0; i < p->n; ++i) {
				if (tmask[p->doc[i]]) continue;
				int relevant = all_pos;
				for (uint32_t a = p->pos_off[i]; !relevant && a < p->pos_off[i + 1]; ++a) relevant = terms[t].opts.field_boost[pt_field(p->fpos[a])] != 0.0f;
				if (relevant) tmask[p->doc[i]] = 1;
			}
		}
		for (uint64_t i = 0; i < total_docs; ++i) mask[i] &= tmask[i];
	}
	for (uint32_t t = 0; t < nterms; ++t) {
		if (terms[t].op != 3) continue;
		for (uint32_t s = 0; s < terms[t].nsub; ++s)
			for (uint64_t i = 0; i < terms[t].subs[s].n; ++i) mask[terms[t].subs[s].doc[i]] = 0;
	}

	/* ---- estimateNumDocsInMerge + the 2-phase gate (mergerimpl.h:487-490) ---- */
	int need_check_removed = 1;
	{
		uint64_t est_or = 0, est_and = UINT64_MAX;
		for (uint32_t t = 0; t < nterms; ++t) {
			if (terms[t].op == 3) continue;
			uint64_t nd = 0;
			for (uint32_t s = 0; s < terms[t].nsub; ++s) nd += terms[t].subs[s].n;
			if (terms[t].op == 2) est_and = nd < est_and ? nd : est_and; else est_or += nd;
		}
		uint64_t est = est_or < est_and ? est_or : est_and;
		if (est > total_docs) est = total_docs;
		uint64_t pop = 0;
		for (uint64_t i = 0; i < total_docs; ++i) pop += mask[i];
		if (est > cfg->merge_limit && total_docs > cfg->merge_limit && pop > cfg->merge_limit) {
			/* preselectMostRelevantDocs */
			uint16_t* score = (uint16_t*)calloc(total_docs, sizeof(uint16_t));
			for (uint32_t t = 0; t < nterms; ++t) {
				if (terms[t].op == 3) continue;
				memset(tmask, 0, total_docs);
				const orc_ft_term_opts* o = &terms[t].opts;
				int same = 1;
				for (uint32_t f = 0; f < nf; ++f) same &= o->field_boost[f] == o->field_boost[0];
				for (uint32_t s = 0; s < terms[t].nsub; ++s) {
					const orc_ft_ppostings* p = &terms[t].subs[s];
					for (uint64_t i = 0; i < p->n; ++i) {
						const uint32_t d = p->doc[i];
						if (!mask[d]) continue;
						float mb = o->field_boost[0];
						if (!same) {
							mb = 0.0f;
							for (uint32_t a = p->pos_off[i]; a < p->pos_off[i + 1]; ++a) {
								const float fb = o->field_boost[pt_field(p->fpos[a])];
								mb = fb > mb ? fb : mb;
							}
						}
						if (mb > 0.0 && !tmask[d]) {
							const float pr = p->proc * mb * o->boost;
							uint16_t p16 = (uint16_t)pr;
							if (p16 > 65535 / 4) p16 = 65535 / 4;
							if (p16 > 65535 - score[d]) p16 = (uint16_t)(65535 - score[d]);
							score[d] = (uint16_t)(score[d] + p16);
							tmask[d] = 1;
						}
					}
				}
			}
			uint64_t* hist = (uint64_t*)calloc(65536, sizeof(uint64_t));
			for (uint64_t i = 0; i < total_docs; ++i) {
				if (!mask[i] || (removed && removed[i])) score[i] = 0;
				hist[score[i]]++;
			}
			uint64_t min_score = 65535, min_score_docs = 0, taken = 0;
			for (uint64_t sc = 65535; sc > 0; --sc) {
				if (taken >= max_merged) break;
				min_score = sc;
				min_score_docs = max_merged - taken;
				taken += hist[sc];
			}
			uint64_t min_taken = 0;
			for (uint64_t i = 0; i < total_docs; ++i) {
				if (!mask[i]) continue;
				if (score[i] > min_score) continue;
				if (score[i] == min_score && min_taken < min_score_docs) {
					++min_taken;
					continue;
				}
				mask[i] = 0;
			}
			free(hist);
			free(score);
			need_check_removed = 0;
			if (out_preselected) *out_preselected = 1;
		}
	}

	/* ---- mergeTerm for every non-NOT term ---- */
	uint32_t* idoff = (uint32_t*)malloc(total_docs * sizeof(uint32_t));
	for (uint64_t i = 0; i < total_docs; ++i) idoff[i] = (uint32_t)max_merged;
	orc_doc_ext* ext = (orc_doc_ext*)calloc(max_merged ? max_merged : 1, sizeof(orc_doc_ext));
	size_t n = 0;
	uint16_t qp = 0;
	for (uint32_t t = 0; t < nterms; ++t) {
		if (terms[t].op == 3) continue;
		++qp;
		for (size_t k = 0; k < n; ++k) {   /* switchToNextWord */
			if (ext[k].nnext) {
				ext[k].last = ext[k].next; ext[k].nlast = ext[k].nnext;
				ext[k].next = NULL; ext[k].nnext = 0;
				ext[k].rank = 0;
			}
		}
		for (uint32_t s = 0; s < terms[t].nsub; ++s) {
			const orc_ft_ppostings* p = &terms[t].subs[s];
			const double idf = orc_bm25_idf(cfg->bm25_type, (double)(total_docs - 1), (double)p->n);
			for (uint64_t i = 0; i < p->n; ++i) {
				const uint32_t d = p->doc[i];
				if (!mask[d]) continue;
				const int added = idoff[d] != (uint32_t)max_merged;
				if (!added && n >= max_merged) continue;
				if (need_check_removed && removed && removed[d]) continue;
				const uint64_t* pos = p->fpos + p->pos_off[i];
				const uint32_t npos = p->pos_off[i + 1] - p->pos_off[i];
				uint8_t field;
				const float rank = rank_from_positions(cfg, &terms[t].opts, idf, p->proc, pos, npos, words + (size_t)d * nf, avg_words, &field);
				if (rank == 0.0f) continue;
				if (!added) {
					out_doc[n] = d; out_proc[n] = rank; out_field[n] = field;
					memset(&ext[n], 0, sizeof(ext[n]));
					ext[n].next = pos; ext[n].nnext = npos; ext[n].rank = rank;
					ext[n].last_term_counted = qp; ext[n].terms_counter = 1;
					idoff[d] = (uint32_t)n;
					++n;
				} else {
					orc_doc_ext* e = &ext[idoff[d]];
					if (e->last_term_counted < qp) { ++e->terms_counter; e->last_term_counted = qp; }
					unsigned dist = orc_positions_distance(e->last, e->nlast, pos, npos);
					if (dist < 1) dist = 1;
					const float norm_dist = orc_bound((float)(1.0 / (double)(float)dist), (float)distance_weight, (float)distance_boost);
					const float final_rank = norm_dist * rank;
					if (final_rank > e->rank) {
						out_proc[idoff[d]] -= e->rank;
						out_proc[idoff[d]] += final_rank;
						e->next = pos; e->nnext = npos;
						e->rank = final_rank;
					}
				}
			}
		}
	}
	/* canBeBoostedByFullMatch: termsCounter == queryParts.size() (NOT parts included); addFullMatchBoost(QueryLength) */
	for (size_t i = 0; i < n; ++i) {
		if (ext[i].terms_counter == nterms && words[(size_t)out_doc[i] * nf + out_field[i]] == (float)nterms)
			out_proc[i] = (float)((double)out_proc[i] * cfg->full_match_boost);
	}
	free(ext);
	free(idoff);
	free(mask);
	free(tmask);
	return orc_ft_post_process(cfg, n, sort_by_rank, out_doc, out_proc, out_field, out_norm);
}
