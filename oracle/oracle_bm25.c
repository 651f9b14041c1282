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
 * PARITY PINNED by the reference's own golden vectors: the `debug_rank()` strings of
 * cpp_src/gtests/tests/unit/ft/ft_generic.cc:326-443 (bm25_norm, position_rank, term_len_boost, term_rank to 7-8 digits) are
 * replayed in tests/test_bm25_oracle.py.  The full ft::Merger does not link standalone (FTConfig's constructor drags the JSON /
 * stop-word / locale TUs and RdxContext drags the activity context), so there is no oracle/_ref build for this path.
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
		const float bm25 = (float)orc_bm25rx_get(idf, cfg->k1, cfg->b, (double)ent_tf[e], (double)words_in_field[f], (double)avg_words[f]);
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
		const double idf = orc_bm25rx_idf((double)(total_docs - 1), (double)p->n);   /* "first doc is always empty" */
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
