// Internal declarations shared by the C-ABI implementation and the kernel translation units.
#pragma once

#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <thread>
#include <mutex>
#include <string>
#include <vector>
#include "ft_packed_decode.h"

namespace rxgpu {

struct ScanParams;

uint32_t scan_grid_x(uint64_t n, int cus);
void launch_scan(int metric, const ScanParams& p, uint32_t nq, uint32_t gridx, hipStream_t s);
struct ScanBf16Params;
bool scan_bf16_supported(uint32_t ld);
void launch_scan_bf16(int metric, const ScanBf16Params& p, uint32_t nq, uint32_t gridx, hipStream_t s);
void launch_filter_approx(const float* approx, uint64_t n, const float* top_dist, const uint32_t* top_count, uint32_t kk, const float* margin,
						  uint32_t* cand_row, uint32_t* cand_cnt, uint32_t cap, uint32_t nq, int cus, hipStream_t s);
void launch_merge(const float* part_dist, const uint32_t* part_row, uint32_t total_per_query, uint32_t kk, uint32_t nq, float* out_dist,
				  uint32_t* out_row, uint32_t* out_count, const uint32_t* gate_cnt, uint32_t gate_cap, hipStream_t s);
void launch_merge_lists(const float* part_dist, const uint32_t* part_row, uint32_t nlists, uint32_t kk, uint32_t nq, float* out_dist, uint32_t* out_row,
						uint32_t* out_count, hipStream_t s);   // the partial results are sorted lists of kk entries: no serial insertions
void launch_merge_shards(const uint32_t* gathered, uint32_t world, uint32_t nq, uint32_t kk, uint32_t shard_rows, float* out_dist,
						 uint32_t* out_row, uint32_t* out_count, hipStream_t s, const uint32_t* slot_base = nullptr, bool sorted = true);
// one shard's HNSW result (unordered, counts <= k) into its [nq][kk] | [nq][kk] slot of the exchange's send buffer, padded with invalid entries
void launch_pack_lists(const float* dist, const uint32_t* row, const uint32_t* count, uint32_t nq, uint32_t k, uint32_t kk, uint32_t* dst_dist, uint32_t* dst_row,
					   hipStream_t s);
void launch_range(int metric, const float* rows, const float* inv_norms, const float* query, uint64_t n, uint32_t stride, uint32_t dim,
				  float radius, int inclusive, float* out_dist, uint32_t* out_row, uint64_t cap, unsigned long long* counter,
				  uint32_t gridx, hipStream_t s);
void launch_range_subset(int metric, const float* rows, const float* inv_norms, const float* query, const uint32_t* ids, uint64_t n_ids,
						 uint32_t stride, uint32_t dim, float radius, int inclusive, float* out_dist, uint32_t* out_row, uint64_t cap,
						 unsigned long long* counter, uint32_t gridx, hipStream_t s);
void launch_distances(int metric, const float* rows, const float* inv_norms, const float* query, uint32_t stride, uint32_t dim,
					  const uint32_t* ids, uint32_t n, float* out, hipStream_t s);

// Pre-filtered search (knn_scan.hip: knn_scan_subset; knn_subset.hip: bitmap -> row list)
uint32_t subset_grid_x(uint64_t n_ids, uint32_t dim, uint32_t kk, int cus);
void launch_scan_subset(int metric, const ScanParams& p, const uint32_t* ids, uint32_t nq, uint32_t gridx, int cus, hipStream_t s);
uint32_t bitmap_tiles(uint64_t n_rows);
void launch_bitmap_count(const uint32_t* words, uint64_t n_rows, uint32_t* tile_scratch, unsigned long long* total, hipStream_t s);
void launch_bitmap_expand(const uint32_t* words, uint64_t n_rows, const uint32_t* tile_scratch, uint32_t* out_rows, uint64_t cap, hipStream_t s);
void launch_ivf_mark_lists(const uint32_t* probe, const uint32_t* probe_cnt, uint32_t nprobe, const uint64_t* list_off, const uint32_t* list_rows,
						   uint32_t* bitmap, hipStream_t s);
void launch_gather_u32(const uint32_t* src, const uint32_t* idx, uint32_t n, uint32_t* out, hipStream_t s);
void launch_check_row_list(const uint32_t* ids, uint64_t n, uint64_t limit, uint32_t* bad, int cus, hipStream_t s);

// Large-k path (k+1 > 64): distance pass + radix select (knn_select.hip).
void launch_all_distances(int metric, const float* rows, const float* inv_norms, const float* query, uint64_t n, uint32_t stride,
						  uint32_t dim, float* out_dist, uint32_t gridx, hipStream_t s);
size_t select_scratch_bytes(uint64_t n);
// Finds the kk smallest (dist,row) of d_dist[0..n) ; writes them UNSORTED to d_out_*; *d_out_n (device) = count (== min(kk,n)).
void launch_select_smallest(const float* d_dist, uint64_t n, uint32_t kk, void* d_scratch, float* d_out_dist, uint32_t* d_out_row,
							hipStream_t s);

// Batched (MFMA) path, knn_batched.hip
struct GemmParams;
size_t gemm_lds_bytes(int mt);
hipError_t launch_gemm(int metric, int mt, int mode, const GemmParams& p, uint32_t grid, hipStream_t s);
void launch_row_stats(const float* rows, const float* inv_norms, uint64_t n, uint32_t stride, uint32_t dim, float* row_sq,
					  unsigned int* stats, int cus, hipStream_t s);
void launch_query_stats(int metric, const float* queries, uint32_t nq, uint32_t mt, uint32_t q_stride, uint32_t dim, const unsigned int* stats,
						float* q_sq, float* margin, bool bf16, hipStream_t s);
struct GemmBf16Params;
hipError_t launch_gemm_bf16(int metric, int mode, int qt, const GemmBf16Params& p, uint32_t grid, hipStream_t s);
void launch_to_bf16(const float* src, uint64_t n, uint32_t stride, uint32_t dim, uint16_t* dst, uint32_t ld, int cus, hipStream_t s, uint64_t first_row = 0,
					bool blocked = false);   // dst = the shadow's base when first_row / blocked are given
void launch_shadow_move(uint16_t* shadow, uint32_t ld, uint64_t from, uint64_t to, bool blocked, hipStream_t s);
void launch_sample_threshold(const float* dense, uint64_t ns, uint32_t nq, uint32_t mt, uint32_t kk, const float* margin, float* thr,
							 hipStream_t s);
void launch_rescore(int metric, const float* rows, const float* inv_norms, const float* queries, uint32_t q_stride, uint32_t stride,
					uint32_t dim, uint32_t nq, uint32_t cap, const uint32_t* cand_cnt, uint32_t* cand_row, float* cand_dist, hipStream_t s);

// HNSW search (hnsw_search.hip)
struct HnswParams;
void launch_hnsw_search(int metric, const HnswParams& p, uint32_t blocks, bool global_cand, hipStream_t s);
struct HnswHelper;
void launch_hnsw_helper(int metric, const HnswParams& p, const HnswHelper& hq, uint32_t groups, hipStream_t s);
struct HnswServer;
// the resident search kernel (one workgroup per mailbox slot); false: this (metric, dim, list size, deleted nodes) has no resident form
bool launch_hnsw_server(int metric, const HnswParams& p, const HnswServer& sv, uint32_t slots, hipStream_t s);
size_t hnsw_server_lds_bytes(const HnswParams& p);
struct HnswPatch;
void launch_hnsw_patch(const HnswPatch& p, uint32_t n_dirty, hipStream_t s);
struct HnswStream;
void launch_hnsw_stream(int metric, const HnswParams& p, const HnswStream& s, uint32_t batch, int mode, bool lds, hipStream_t st);
struct HnswRange;
void launch_hnsw_range(int metric, const HnswParams& p, const HnswRange& r, hipStream_t s);

// Hybrid FT + KNN rank fusion on the device (hybrid_fuse.hip)
constexpr int kMaxFuseKnn = 1024;      // KNN entries one fusion takes (k of the KNN condition)
// what hybrid_prepare_kernel leaves in HBM for hybrid_join_kernel, next to the FT documents in their final mutual order
struct HybridFuseState {
	uint32_t n_valid;            // documents that passed postProcessResults = entries of the prepared list
	uint32_t max_id;
	uint32_t result_in_second;   // which half of the scratch arrays holds the prepared list
	uint32_t pad;
	uint32_t cls_pos[256];       // RRF position of a rank class
	uint32_t cls_key[256];       // fused rank of an FT-only document of the class, as an order key (smaller = earlier)
	float cls_rank[256];         // ... and as the float that is returned
	uint32_t cls_group[256];     // classes with equal fused ranks share a group
	uint32_t grp_key[256];       // key of group g (0xFFFFFFFF: unused)
	uint32_t grp_start[260];     // first prepared position of group g; [256] = n_valid
};
struct HybridFuseArgs {
	// FT side.  Either the merge train's raw output (ft_doc + ft_proc: postProcessResults is applied here with min_rank) or documents with
	// their uint8 ranks (ft_doc + ft_rank_u8).  Ids unique, any order.  ft_count_ptr (device) overrides ft_n when set.
	const uint32_t* ft_doc;
	const float* ft_proc;
	const uint8_t* ft_rank_u8;
	const uint32_t* ft_count_ptr;
	const uint16_t* ft_terms;          // optional (raw merge output of a query with multi-word synonyms): 0xFFFF marks a document that holds only
	                                   // parts of a synonym — removed before postProcessResults (mergerimpl.h:533-555), i.e. absent here
	uint32_t ft_n, ft_cap;             // ft_cap: what the scratch arrays hold
	float min_rank;
	const int32_t* row_of_doc;         // vdoc -> row id (1:1), or null: the document number is the row id
	// KNN side: the search's (dist, row) list, best first
	const float* knn_dist;
	const uint32_t* knn_row;
	const uint32_t* knn_count_ptr;     // device count of valid entries, or null: knn_n
	uint32_t knn_n, k;                 // entries present / entries that take part (the list holds k + 1 when the caller checks boundary ties)
	int32_t knn_negate;                // IP / cosine: the planner's rank is -distance (hnsw_index.cc:261-270)
	int32_t metric_l2;                 // runs of equal KNN ranks: L2 ascends, IP / cosine descend
	const int32_t* rowid_of_row;       // internal row -> row id, or null: identity
	// reranker: kind 0 RRF (params[0] = rank_const), 1 linear (kKnn, knnDefault, kFt, ftDefault, c)
	int32_t kind, is_union, desc;
	double params[5];
	int32_t* out_ids;
	float* out_ranks;
	uint32_t* out_header;              // [0] count, [1] flags (1: distance tie at the k-th place), [2] head size, [3] tail size
	unsigned long long* dbg;           // phase stamps of the join kernel (100 MHz wall clock), 8 words, or null
	HybridFuseState* state;
	uint32_t* scratch_key;             // [2 * ft_cap]
	uint16_t* scratch_cls;             // [2 * ft_cap]
};
hipError_t launch_hybrid_prepare(const HybridFuseArgs& a, hipStream_t st);   // FT only: may run while the KNN search is still going
hipError_t launch_hybrid_join(const HybridFuseArgs& a, hipStream_t st);      // needs both halves

// ft_fast merge (ft_merge.hip): Merger::mergeSimple / mergeTerm + restricting bitmask + preselect, restated ORDER-FREE so that a whole
// query is a fixed number of launches (see the header of ft_merge.hip)
struct FtPosSubterm {
	uint64_t n;
	const uint32_t* doc;
	const uint32_t* ent_off;
	const uint8_t* ent_field;
	const uint32_t* ent_tf;
	const uint32_t* ent_first_pos;
	const uint32_t* pos_off;   // [n + 1]; null for words uploaded without positions (single-term merge only)
	const uint64_t* fpos;      // PosType words (idrelset.h:14-32): pos | arrayIdx << 28 | field << 56
	double idf;                // the calculator's IDF for this sub-term (bm25.h), computed on the host
	float proc;
	uint32_t term;             // query term index -> FtPlan::terms
	uint16_t qp;               // 1-based index among the terms that are not NOT (mergeTerm's qpIdx); 0 for a NOT term
	uint16_t ord_in_term;      // position inside its term (SortSubterms order)
	uint32_t row;              // row of the per-slot entry table = index among the merged (non-NOT, non-empty) sub-terms
	const uint32_t* range_off; // [n_ranges + 1]: first posting with doc >= k * kFtRangeDocs (built when the word is uploaded)
	uint32_t n_ranges;
	// A row of a merged PHRASE (ft_phrase.hip): the documents one sub-term of the phrase's first term added to the PhraseMerger, with their
	// phrase rank / field (what mergePhrase reads, mergerimpl.h:39-90) instead of entries; fpos = lastPhrasePositions.  Null for a word.
	const float* pre_rank;
	const uint8_t* pre_field;
	uint16_t prev_term_qp;     // phrase row: qp of the last plain term in front of the phrase (0: none) — the last switchToNextWord before it
	uint8_t phrase;            // 1: phrase row
	uint8_t suppressed;        // SubtermResults::Suppressed() (querymergedata.h:32, set by SupressDuplicatesInSynonyms :221-241): the sub-term of a
	                           // multi-word synonym is a word the query's own terms found already — its postings only count terms (mergerimpl.h:144-151)
};
constexpr uint32_t kFtSuppressedRank = 0x7FC00001u;   // rank bits of a suppressed sub-term's record (a NaN: never a rank, never 0)
// (qp, phrase flag, prev_term_qp) of a row as the replay carries it
__host__ __device__ inline uint32_t ft_row_qpw(const FtPosSubterm& s) { return uint32_t(s.qp) | (uint32_t(s.phrase) << 15) | (uint32_t(s.prev_term_qp) << 16); }
struct FtTermCfg {             // what calcTermRank reads: FTConfig + the FtDslOpts of ONE query term
	uint32_t num_fields;
	int32_t bm25_type;         // kFtBm25Rx / Classic / WordCount (ft_rank.hip.h)
	const float* words;
	const float* avg_words;
	double k1, b, summation_ratio;
	float opts_boost, term_len_boost_in;
	const float* field_boost;
	const uint8_t* need_sum_rank;
	const float *bm25_boost, *bm25_weight, *term_len_boost, *term_len_weight, *position_boost, *position_weight;
	int32_t op;                // 1 OR, 2 AND, 3 NOT
	uint32_t sub_begin, sub_end; // the term's (non-empty) sub-terms in FtPlan::subs, in SortSubterms order
	uint8_t same_boost;        // every field has the same boost (calcTermScores' shortcut)
	uint8_t all_pos_boost;     // every field has a non-zero boost (calcTermBitmask's shortcut)
	uint8_t phrase;            // the part is a phrase: its sub-terms are the rows ft_phrase.hip produced; field_boost = ones
	uint8_t pad0;
	uint32_t phrase_proc16;    // PhraseResults::CalcProc16 (querymergedata.h:117-127): what GetMergedDocsScore adds (phrasemerger.h:326-333)
	// an AND part with multi-word synonyms (PhraseOrTerm::SynonymsIds): the documents that hold EVERY term of one of them, as a bitmap over
	// the documents — OR-ed into the part's term mask before it restricts (buildRestrictingBitmask, mergerimpl.h:347-361); ft_syn_masks
	const uint32_t* syn_mask;
};
// Multi-word synonyms (QueryMergeData::synonyms, querymergedata.h:178-192): their terms follow the query parts in FtPlan::terms (op = OR for
// the pre-score pass: calcTermScores counts them like any term, mergerimpl.h:393-397, and they never restrict on their own)
struct FtSynonym {
	uint32_t term_begin, term_end;   // its terms in FtPlan::terms
	uint32_t end_qp;                 // qp of its last term (every term takes a qp, NOT terms too: mergerimpl.h:511-514)
	uint32_t nterms;                 // Synonym::NumTerms()
};
struct FtSynMaskJob {                // one AND part's synonym mask
	uint32_t syn_begin, syn_end;     // into FtPlan::job_syns
	uint32_t* out;                   // [nwords]
};
struct FtGridEntry {           // block range of one sub-term in a posting-side grid (blocks of kFtBlockPostings postings)
	uint32_t block_base;
	uint32_t sub;              // index into FtPlan::subs
};
constexpr int kFtPassItems = 4;            // postings per thread in the posting-side kernels
constexpr uint32_t kFtRangeDocs = 8192;    // documents per workgroup of the document-range kernel (ft_ranges); multiple of 32
constexpr int kFtBlockPostings = 256 * kFtPassItems;
inline uint32_t ft_pass_blocks(uint64_t n) { return uint32_t((n + kFtBlockPostings - 1) / kFtBlockPostings); }

// Everything one merge needs on the device.  Pointers into per-index scratch; scalar members by value (the struct travels as a kernel argument).
struct FtPlan {
	const FtPosSubterm* subs;
	const FtTermCfg* terms;
	const FtGridEntry* merge_grid;   // merged sub-terms in (term, sub-term) order
	uint32_t n_merge_entries, merge_blocks;
	uint32_t nterms, n_rows, n_subs;   // nterms = entries of `terms`: the query parts (a phrase is one part), then the synonyms' terms
	uint32_t n_parts;                  // queryParts.size()
	uint32_t n_part_qp;                // qp of the last merged query part: documents created behind it are the synonyms' (mergerimpl.h:510)
	const FtSynonym* syns;             // [n_syn]
	uint32_t n_syn;
	const FtSynMaskJob* syn_jobs;      // [n_syn_jobs] -> ft_syn_masks
	const uint32_t* job_syns;
	uint32_t n_syn_jobs;
	uint32_t query_len;                // QueryMergeData::QueryLength(): the terms inside phrases counted one by one (addFullMatchBoost)
	uint64_t total_docs, nwords;
	uint32_t max_merged, merge_limit;
	uint8_t simple;            // Merger::mergeSimple (one term): max over sub-terms, first maximum wins; no positions
	uint8_t prescore;          // the host-side half of the 2-phase gate held: pre-scores are collected, the device decides on popcount
	uint8_t check_removed;
	float distance_weight, distance_boost;
	double full_match_boost;   // FTConfig::fullMatchBoost, applied by ft_replay (addFullMatchBoost)
	const uint8_t* removed;
	const uint8_t* excluded;
	uint32_t* mask;            // restrictingMask_ [nwords]
	uint16_t* score;           // [total_docs]
	// pre-score histogram in kFtHistCopies interleaved copies (workgroup b of ft_ranges adds to copy b % kFtHistCopies; the reader sums them):
	// every workgroup holds the same handful of scores, and same-address device atomics are served one at a time.
	// Copy c = hist + c * kFtHistStride: [65536] documents per pre-score, then [1024] documents per chunk of 64 scores.
	uint32_t* hist;
	// admission (ft_rank_all -> ft_adders -> ft_finish): the eligible postings with a non-zero rank, bucketed by document range
	uint4* b_rec;              // [merged postings] records {doc, posting index, rank bits, row | field << 16}; bucket r starts at bucket_off[r]
	uint32_t* bucket_off;      // [n_ranges] = merged postings in front of the range (sum of the sub-terms' range offsets), written by ft_ranges
	uint32_t* bucket_cnt;      // [n_ranges] records in the bucket; kept zero between merges (ft_finish clears its own)
	uint32_t* adders;          // [n_rows][n_ranges]: documents first met in (sub-term row, range); then its exclusive prefix in place = slot bases
	uint32_t n_ranges;
	float* e_rank;             // per-slot entry table [n_rows][max_merged]: 0 = the document has no posting in that sub-term
	uint32_t* e_idx;
	uint8_t* e_field;
	unsigned long long* dbg;   // RXGPU_FT_STAMPS=<workgroup>: wall-clock stamps of that workgroup's phases (null otherwise)
	uint32_t dbg_block;
	uint32_t* sync;            // kFtSync* words; kept zero between merges (the last workgroup of ft_finish clears them)
	unsigned long long* lookback_pre;     // [ceil(nwords / (256 * 4))]
	// packed result: header (4 x u32: numDocs, error flag, preselected, 0) then doc[max_merged] u32, proc[max_merged] f32,
	// terms_counter[max_merged] u16, field[max_merged] u8 — one D2H copy
	uint32_t* out_header;
	void* host_out;            // device-visible address of the caller's pinned staging buffer: ft_export copies the used part of the result there
	uint32_t* out_doc;
	float* out_proc;
	uint16_t* out_terms_counter;
	uint8_t* out_field;
	// MergeDataAreas<Area> (merger.h:39-41, 182-204; areaholder.h:56-155): the highlight / snippet areas of every merged document, built by the
	// replay from the positions of the document's postings in merge order.  max_areas = FTConfig::maxAreasInDoc (> 0), 0: no areas.
	// area_hdr [max_merged][area_fields][2] = {entries held (AreasInField::data_.size()), insertions so far (index_)}, zeroed by the host in
	// front of the train; out_areas [max_merged][area_fields][max_areas][3] = {start, end, arrayIdx} in data_ order.
	uint32_t max_areas, area_fields;
	uint32_t* area_hdr;
	uint32_t* out_areas;
	// Document-range shards (SURVEY 8e "BM25", rxgpu_ft_sharded.hip): this handle merges the documents of the ranges [range_begin,
	// range_begin + range_count) only — its posting lists hold just those documents' fragments (global ids; idf from the global N / df) — and
	// the three per-query facts that span the shards travel between the kernels: the pre-score histogram + mask popcount (behind ft_ranges;
	// shard_hist = every shard's folded histogram, [n_shards][kFtFoldWords]: the sum is the threshold's input, the entries of the shards in
	// front give the tie quota already used at the threshold score), and the table of ft_adders (every shard fills its own columns, the
	// sum is the global table the slot bases come from).  range_count == 0: the whole index (unsharded).
	uint32_t range_begin, range_count;
	uint32_t shard_index, n_shards;
	const uint32_t* shard_hist;   // gathered layout: shard s at [shard_pos[s]][kFtFoldWords]
	const uint32_t* shard_pos;    // [n_shards]
	// The train for SPARSELY hit document ranges (ft_sparse.hip): one wavefront per (query, range), bitmaps of the range's documents per
	// sub-term in LDS, nothing per document in HBM.  sparse = 1: the host found the query eligible (ft_sparse_eligible, rxgpu_ft_capi.hip).
	uint8_t sparse;
	uint8_t sp_empty_and;         // an AND term without postings: no document passes the mask
	const uint32_t* removed_bits; // [nwords] DocRemoved as one bit per document (null: none removed); built by rxgpu_ft_set_docs
	const uint32_t* excluded_bits;// [nwords] docsExcluded of this merge as bits (null: none)
	unsigned long long* lb_units; // [n_ranges] look-back words of ft_sp_select's ordered tie count; kept zero between merges
	// the merged documents as ft_sp_select / ft_sp_place found them, one task each, replayed by ft_sp_replay:
	//   t_doc [max_merged] document; t_pos [max_merged] its slot | 1 << 31, or merge row << 24 | rank among the (row, range)'s documents;
	//   t_idx [max_merged][n_rows] posting index + 1 of the document in sub-term row r, 0: none
	uint32_t* t_doc;
	uint32_t* t_pos;
	uint32_t* t_idx;
	struct SpSub {                // what the unit kernels read of sub-term si (lane si loads entry si)
		const uint32_t* doc;
		const uint32_t* range_off;
		uint32_t n, n_ranges;
		uint32_t attr;            // proc16 | first sub-term of its term << 16 | AND term << 17 | NOT term << 18 | merge row << 20
		uint32_t pad;
	} sp_sub[16];
};
static_assert(sizeof(FtPlan::SpSub) == 32, "one 32-byte entry per lane");
constexpr uint32_t kFtFoldWords = 65536 + 1024 + 64;   // one shard's folded histogram (fine + chunk counters), [65536 + 1024] = its mask popcount
enum : uint32_t { kFtSyncError = 0, kFtSyncPop = 1, kFtSyncPreTicket = 4, kFtSyncNumDocs = 6, kFtSyncDoneFinish = 9,
				  // the sparse train: threshold score / documents kept at it / flags (bit 0 preselect on, bit 1 every tie is kept), the ticket of ft_sp_select
				  kFtSyncThrScore = 10, kFtSyncThrDocs = 11, kFtSyncThrFlags = 12, kFtSyncSpTicket = 13, kFtSyncTasks = 14, kFtSyncWords = 16 };
constexpr uint32_t kFtHistCopies = 8, kFtHistStride = 65536 + 1024;
constexpr uint32_t kFtRangeShift = 13;     // log2(kFtRangeDocs)
static_assert((1u << kFtRangeShift) == kFtRangeDocs, "document ranges are powers of two");
// Q merges over one index in ONE train (grid.y = query; a single merge is a batch of one): the plans in HBM + their host copy
hipError_t launch_ft_merge(const FtPlan* plans, const FtPlan* host_plans, uint32_t nq, hipStream_t st);
// the same train in the three pieces a sharded merge exchanges between: 0 = [syn masks] + ft_ranges, 1 = [ft_preselect_apply] + ft_rank_all +
// ft_adders, 2 = [ft_slot_bases] + ft_finish
hipError_t launch_ft_merge_phase(const FtPlan* plans, const FtPlan* host_plans, uint32_t nq, int phase, hipStream_t st);
// the train for sparsely hit ranges (ft_sparse.hip) over nq plans that all have sparse = 1, those with prescore = 1 in front: ft_sp_scan,
// [ft_sp_threshold, ft_sp_select], ft_slot_bases, [ft_sp_place], ft_sp_replay
hipError_t launch_ft_merge_sparse(const FtPlan* plans, const FtPlan* host_plans, uint32_t nq, hipStream_t st);
void launch_ft_slot_bases(const FtPlan* plans, uint32_t nq, hipStream_t st);   // ft_merge.hip
constexpr uint32_t kFtSparseSubs = 16;   // sub-terms (NOT terms' included) a sparse merge holds bitmaps for
void launch_ft_shard_fold(const FtPlan* plan, uint32_t* dst, hipStream_t st);                                            // hist copies + popcount -> dst [kFtFoldWords]
void launch_ft_shard_hist_combine(const FtPlan* plan, const uint32_t* gathered, const uint32_t* pos, uint32_t n_shards, hipStream_t st);   // gathered [..][kFtFoldWords]
void launch_ft_shard_table_sum(uint32_t* table, const uint32_t* gathered, const uint32_t* pos, uint32_t n_shards, uint64_t n, uint64_t stride, hipStream_t st);   // gathered [..][stride]
constexpr uint32_t kFtBatchMax = 64;
struct FtImportBatch {             // pieces of one batched upload (pinned staging -> HBM), 16-byte words
	const void* src[kFtBatchMax + 1];
	void* dst[kFtBatchMax + 1];
	uint32_t n16[kFtBatchMax + 1];
	uint32_t n;
};
hipError_t launch_ft_import_batch(const FtImportBatch& b, hipStream_t st);

// PhraseMerger::Merge (phrasemergerimpl.h:161-329) for ONE phrase, ft_phrase.hip.  Three launches: admission over the first term's
// postings (ordered prefix = mergeData_ order), one thread per admitted document through the terms, packing of the documents with a
// non-zero rank into one posting list per sub-term of the first term (the rows the main merge then treats like words).
struct FtPhrasePlan {
	const FtPosSubterm* subs;      // the sub-terms of the phrase's terms, term after term, SortSubterms order
	const FtTermCfg* terms;        // [nterms]: calcTermRank configuration, sub_begin / sub_end into subs
	const int32_t* distance;       // [nterms] FtDslOpts::distance of the term (MergeWithDist's dist)
	const FtGridEntry* grid;       // the first term's sub-terms in blocks of kFtBlockPostings postings
	uint32_t nterms, n_grid, grid_blocks, n_rows0;   // n_rows0: sub-terms of the first term
	uint32_t max_merged;           // min(mergeLimit, Term(0).MaxVDocs()) (phrasemerger.h:341-342)
	uint32_t n_ranges;
	uint64_t total_docs;
	float distance_weight, distance_boost;
	const uint8_t* removed;
	const uint8_t* excluded;
	unsigned long long* lookback;  // [grid_blocks], zero
	uint32_t* sync;                // [8], zero: 0 ticket, 1 error, 2 admitted, 3 alive, 4-5 sum of caps (u64), 6-7 workspace top (u64)
	uint32_t* slot_doc;            // [max_merged] per admitted document, mergeData_ order
	uint32_t* slot_row;            // sub-term of the first term that added it
	uint32_t* slot_cap;            // positions the document can carry from one term to the next
	float* slot_proc;
	uint8_t* slot_field;
	uint64_t* slot_pos;            // offset of lastPhrasePositions in the workspace
	uint32_t* slot_npos;
	uint64_t* ws;                  // position workspace (2 x sum of caps)
	// packed rows: row r = [row_base[r], row_base[r] + row_cnt[r]) of the arrays below; pos_off holds row_cnt + 1 entries per row
	uint32_t* row_cnt;             // [n_rows0]
	uint32_t* row_base;            // [n_rows0]
	uint32_t* out_doc;
	float* out_rank;
	uint8_t* out_field;
	uint32_t* out_pos_off;
	uint64_t* out_fpos;
	uint32_t* out_range_off;       // [n_rows0][n_ranges + 1]
	uint32_t* out_header;          // [4 + n_rows0] (pinned, device view): admitted, error, alive, positions, then row_cnt
};
constexpr uint32_t kFtPhraseRowPad = 64;   // entries: every packed row starts on a 256-byte boundary
hipError_t launch_ft_phrase_admit(const FtPhrasePlan& p, hipStream_t st);
hipError_t launch_ft_phrase_docs(const FtPhrasePlan& p, uint32_t admitted, hipStream_t st);
hipError_t launch_ft_phrase_pack(const FtPhrasePlan& p, hipStream_t st);
hipError_t launch_ft_export(const FtPlan* plans, const FtPlan* host_plans, uint32_t nq, hipStream_t st);
hipError_t launch_ft_import(const void* host_plan_device_view, void* dev_plan, size_t bytes, hipStream_t st);
// ft_packed.hip: PackedIdRelVec streams -> flat posting arrays, one thread per word (counting pass, then writing pass)
// pieces of the packed streams for the wavefront decoder (null: the one-thread-per-word kernels)
struct FtPackedSegs {
	const uint32_t* seg_word;    // [nsegs] word (launch order) of a piece
	const uint32_t* seg_first;   // [nwords + 1] first piece of a word
	FtPackedCheckpoint* cps;     // [nsegs], byte_off preset to ~0
	uint32_t nsegs;
};
hipError_t launch_ft_packed_count(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords, uint32_t num_fields,
								   FtPackedCounts* counts, const FtPackedSegs* segs, hipStream_t st, uint32_t first_word, uint32_t word_count);
hipError_t launch_ft_packed_write(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords, uint32_t num_fields,
								   const FtPackedOut* outs, FtPackedCounts* counts, const FtPackedSegs* segs, hipStream_t st);

void set_error(const std::string& msg);

}  // namespace rxgpu

// A growable device buffer.
struct rxgpu_devbuf {
	void* ptr = nullptr;
	size_t bytes = 0;
	int ensure(size_t need);
	void release();
};

// Per-search scratch: one is checked out per host-side search call (or bound to a caller stream).
struct rxgpu_search_ctx {
	hipStream_t stream = nullptr;
	bool own_stream = false;
	rxgpu_devbuf d_queries, d_part_dist, d_part_row, d_out_dist, d_out_row, d_out_count, d_misc, d_select;
	rxgpu_devbuf d_qpad, d_qstats, d_dense, d_cand_row, d_cand_dist, d_cand_cnt;   // batched path
	rxgpu_devbuf d_visited, d_gcand_d, d_redo;                                     // HNSW (d_gcand_d: (dist bits, id) entries)
	rxgpu_devbuf d_helper, d_helper_bits;                                          // HNSW: overflow queue of a batch, bitsets of its helper workgroups
	rxgpu_devbuf d_top;                                                            // bf16-pruned scan: approximate top lists
	rxgpu_devbuf d_ivf;                                                            // IVF: the coarse search's lists, distances, count
	rxgpu_devbuf d_subset, d_bitmap, d_tiles;                                      // pre-filtered search: row list, allowed-rows bitmap, tile sums
	void* h_pinned = nullptr;
	size_t h_pinned_bytes = 0;
	// second stream + events (created on first use): work that does not depend on the query upload — zeroing the visited bitsets of an
	// HNSW launch — runs beside it
	hipStream_t aux_stream = nullptr;
	hipStream_t aux2_stream = nullptr;   // second half of a large HNSW batch: its upload runs beside the first half's searches
	hipEvent_t split_done = nullptr;
	hipEvent_t aux_done = nullptr, main_done = nullptr;
	int ensure_aux();
	int ensure_pinned(size_t need);
	void release();
};

struct rxgpu_profile_slot {
	std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};

struct rxgpu_index;
namespace rxgpu {
struct ShardSet;
void sharded_destroy(struct ::rxgpu_index* h);
uint64_t sharded_device_bytes(const struct ::rxgpu_index* h);
int sharded_upload_rows(struct ::rxgpu_index* h, uint64_t first_row, uint64_t n, const float* rows, const float* inv_norms);
int sharded_truncate(struct ::rxgpu_index* h, uint64_t count);
int sharded_move_row(struct ::rxgpu_index* h, uint64_t from, uint64_t to);
int sharded_search_knn_impl(struct ::rxgpu_index* h, const float* queries, uint32_t nq, uint32_t kk, const uint32_t* row_ids, uint64_t n_ids, float* out_dist,
							uint32_t* out_row, uint32_t* out_count);
int sharded_search_range_impl(struct ::rxgpu_index* h, const float* query, float radius, int inclusive, const uint32_t* row_ids, uint64_t n_ids,
							  float* out_dist, uint32_t* out_row, uint64_t cap, uint64_t* out_total);
int sharded_distances(struct ::rxgpu_index* h, const float* query, const uint32_t* rows, uint32_t n, float* out_dist);
// Sharded HNSW: a graph per shard (attached through rxgpu_index_shard handles), SearchKnn fans out and meets in the exchange
struct HnswSink {
	uint32_t* d_dist;   // [nq][kk] distance bits   (device memory of the shard's GPU)
	uint32_t* d_row;    // [nq][kk] shard-local rows, kInvalidRow past a query's count
	uint32_t kk;
};
// queries: float rows, or (qcorr != null) SQ8 codes with their corrective offsets and normCoefs — the shard then searches its code table
int hnsw_search_to_sink(struct ::rxgpu_index* shard, const void* queries, const float* qcorr, const float* qnorm, uint32_t nq, uint32_t k, uint32_t ef, const HnswSink& sink);
int sharded_hnsw_search_knn(struct ::rxgpu_index* h, const void* queries, const float* qcorr, const float* qnorm, uint32_t nq, uint32_t k, uint32_t ef, float* out_dist,
							uint32_t* out_row, uint32_t* out_count);
int sharded_hnsw_search_range(struct ::rxgpu_index* h, const float* query, float radius, uint32_t ef, float* out_dist, uint32_t* out_row, uint64_t cap,
							  uint64_t* out_total);
}  // namespace rxgpu

namespace rxgpu {
// The resident HNSW search kernel of an index and its mailbox (rxgpu_hnsw_server.hip)
struct HnswServerState;
struct HnswServerConfig {
	uint32_t slots = 256;      // RXGPU_HNSW_SERVER_SLOTS: workgroups = requests in flight (one per CU of an MI355X; an idle slot looks at its mailbox word every ~7 us)
	uint32_t idle_us = 2000;   // RXGPU_HNSW_SERVER_IDLE_US: the kernel leaves after so long without a request
	bool nbl = false;          // RXGPU_HNSW_NBL=1: link blocks come along with a hop's rows (read when the mailbox is made; off by default)
	bool spec = false;         // RXGPU_HNSW_SPEC=1: look-ahead distance batches (read when the index's mailbox is made; off by default)
	uint32_t life_ms = 50;     // RXGPU_HNSW_SERVER_LIFE_MS: ... and after so long in any case (the next caller launches the next one)
};
// 1: served, 0: not served (the caller takes the launches), < 0: -(RXGPU error code is returned as is by the caller) — see the .hip
int hnsw_server_search(struct ::rxgpu_index* h, const HnswServerConfig& cfg, const float* query, uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row,
					   uint32_t* out_count);
void hnsw_server_quiesce(struct ::rxgpu_index* h);      // before the index changes: the resident kernel leaves, none is queued
void hnsw_server_destroy(struct ::rxgpu_index* h);
void hnsw_servers_pause_device(int device);             // before a device-wide wait: every index's resident kernel on that device leaves
extern std::atomic<int> g_resident_kernels;             // resident search kernels that may be alive: frees are deferred meanwhile (rxgpu_capi.hip)
void free_or_retire(void* ptr, size_t bytes, bool host);
void drain_retired();
hipError_t device_wait_all(int device);                 // hipDeviceSynchronize behind hnsw_servers_pause_device
void hnsw_server_times(const struct ::rxgpu_index* h, uint64_t* device_us, uint64_t* caller_us);
void hnsw_server_counters(const struct ::rxgpu_index* h, uint64_t* served, uint64_t* generations);
struct DeviceGuardLite {
	int prev = -1;
	explicit DeviceGuardLite(int dev) {
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		if (prev != dev) (void)hipSetDevice(dev);
	}
	~DeviceGuardLite() {
		if (prev >= 0) (void)hipSetDevice(prev);
	}
};
}  // namespace rxgpu

struct rxgpu_index {
	rxgpu::ShardSet* shard_set = nullptr;   // non-null: a row-range sharded index (rxgpu_sharded.hip); the fields below describe the whole
	int metric = 0;
	uint32_t dim = 0;
	uint32_t stride = 0;     // floats per row in HBM
	uint64_t capacity = 0;   // rows allocated (owned storage only)
	uint64_t count = 0;
	int device = 0;
	int cus = 256;

	float* d_rows = nullptr;       // owned or adopted
	float* d_inv_norms = nullptr;  // cosine only
	bool adopted = false;

	// batched path: per-row |x|^2 (L2) and the maxima entering the rounding bound; recomputed lazily after mutations
	float* d_row_sq = nullptr;
	uint64_t row_sq_capacity = 0;
	unsigned int* d_stats = nullptr;
	bool stats_valid = false;
	uint16_t* d_rows_bf16 = nullptr;   // bf16 shadow of the rows for the nomination GEMM (built lazily with the row statistics)
	uint64_t bf16_capacity = 0;
	bool bf16_blocked = true;   // layout of the shadow (knn_kernels.hip.h); RXGPU_SHADOW_BLOCKED=0 when the shadow is first built: row-major (A/B)
	bool bf16_valid = false;
	bool bf16_unavailable = false;     // the shadow did not fit in HBM: nominate on the f32 rows instead (still exact, still on the GPU)

	// HNSW graph mirror (rxgpu_hnsw_attach_graph)
	uint32_t* d_links0 = nullptr;
	uint64_t* d_upper_off = nullptr;
	uint32_t* d_upper = nullptr;
	uint8_t* d_deleted = nullptr;
	// IVF: inverted lists over this index's rows as CSR (rxgpu_index_set_lists)
	uint64_t* d_list_off = nullptr;   // [nlist + 1]
	uint32_t* d_list_rows = nullptr;  // [lists_rows]
	uint32_t nlist = 0;
	uint64_t lists_rows = 0, lists_count = 0;   // rows listed; the index row count the lists were built for
	// SQ8 copy of the rows (rxgpu_hnsw_attach_sq8): codes [sq8_n][dim], stored corrective offsets, alpha^2
	uint8_t* d_codes = nullptr;
	float* d_corr = nullptr;
	float sq8_alpha2 = 0.f;
	uint64_t sq8_n = 0, sq8_cap = 0;   // rows with codes / rows the tables are allocated for
	uint64_t graph_n = 0, graph_deleted = 0;
	uint64_t graph_rows_cap = 0, graph_upper_cap = 0, graph_upper_used = 0;   // allocated level-0 rows / upper blocks (rxgpu_hnsw_patch_graph grows in place)
	uint32_t graph_M = 0, graph_maxM0 = 0;
	int graph_maxlevel = -1;
	uint32_t graph_entry = 0;
	bool graph_attached = false;
	unsigned long long* d_hnsw_stats = nullptr;
	rxgpu::HnswServerState* hnsw_server[2] = {nullptr, nullptr};   // the resident search kernels' mailboxes: [0] ef <= 128, [1] ef <= 256 (made at the first single query of the class)
	bool hnsw_server_failed[2] = {false, false};
	std::atomic<uint64_t> hnsw_lds_reruns{0};   // searches whose candidate heap outgrew its first LDS area and were re-run with the largest one
	std::atomic<uint64_t> hnsw_tie_reruns{0};   // queries the sorted-list search handed to the heap kernel (equal distances met)

	std::mutex mtx;  // guards ctx pool + profile state
	std::vector<rxgpu_search_ctx*> free_ctx;
	std::map<void*, rxgpu_search_ctx*> stream_ctx;
	int32_t* d_row_ids = nullptr;   // internal row -> row id (label >> 32) for consumers on the device (hybrid fusion); null until uploaded
	uint64_t row_ids_cap = 0;
	// rxgpu_search_knn_resident: the result stays in the context's buffers for a consumer on the device.  One context per CALLING THREAD:
	// the list a thread left in HBM lives until that thread's next resident search, whatever other threads search meanwhile.
	// A thread that ends gives its context back to the pool (rxgpu_capi.hip: ResidentThread), found through `serial` in the table of live indexes.
	std::map<std::thread::id, rxgpu_search_ctx*> resident_ctx;
	std::mutex resident_mtx;
	uint64_t serial = 0;

	bool profiling = false;
	std::map<std::string, rxgpu_profile_slot> profile;
};
