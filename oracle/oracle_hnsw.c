/* TEST INFRASTRUCTURE ONLY — the CHECKER for the GPU HNSW search, never the product.
 *
 * Plain-C restatement of the reference's HNSW *search* (Restream/reindexer v5.15.0,
 * cpp_src/core/index/float_vector/hnswlib/hnswalg.h) over a flat export of the graph:
 *   getLayer0EntryPoint            hnswalg.h:799-827
 *   initLayer0SearchState          hnswalg.h:829-858
 *   layer0ShouldStopBeforePop      hnswalg.h:860-869
 *   runLayer0Step                  hnswalg.h:871-963   (non-streaming branch)
 *   searchBaseLayerST / search     hnswalg.h:966-975, 1978-1985
 *   SearchKnn                      hnswalg.h:1988-2012
 * The two working heaps are the reference's PriorityQueue (priority_queue.h:7-152) with CompareByFirst
 * (hnswalg.h:581-585): ties are resolved by the sift mechanics, which are reproduced step for step, so on the same
 * graph this returns exactly what the reference returns.
 *
 * PARITY PINNED: tests/test_oracle_vs_ref.py::test_hnsw_search_* compare it with the real engine (oracle/_ref) on graphs
 * built by the real engine (exported by oracle/ref/ref_shim.cc) and tests/golden/hnsw.npz holds such a graph + results.
 *
 * Flat graph (produced by ref_hnsw_export_* and by the product's host builder):
 *   links0   u32 [n][1+maxM0]   slot 0 = neighbour count
 *   upper    u32 [blocks][1+M]  node i owns levels[i] consecutive blocks starting at upper_off[i] (level 1 first)
 *   levels   i32 [n], labels u64 [n], deleted u8 [n], vectors f32 [n][dim], inv_norms f32 [n] (cosine only)
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_knn.h"

typedef struct {
	float d;
	uint32_t id;
} hpair;

/* PriorityQueue<pair<float,tableint>, vector, CompareByFirst>: max-heap on .d only */
typedef struct {
	hpair* c;
	size_t n, cap;
} fheap;

static void fh_reserve(fheap* h, size_t need) {
	if (need > h->cap) {
		h->cap = h->cap ? h->cap * 2 : 256;
		if (h->cap < need) h->cap = need;
		h->c = (hpair*)realloc(h->c, h->cap * sizeof(hpair));
	}
}
static void fh_sift_up(fheap* h, size_t child) { /* priority_queue.h:109-123 */
	const hpair v = h->c[child];
	while (child > 0) {
		const size_t parent = (child - 1) / 2;
		if (!(h->c[parent].d < v.d)) break;
		h->c[child] = h->c[parent];
		child = parent;
	}
	h->c[child] = v;
}
static void fh_sift_down(fheap* h, size_t parent, size_t size) { /* priority_queue.h:125-151 */
	const hpair v = h->c[parent];
	for (;;) {
		const size_t left = parent * 2 + 1;
		if (left >= size) break;
		size_t best = left;
		const size_t right = left + 1;
		if (right < size && h->c[left].d < h->c[right].d) best = right;
		if (!(v.d < h->c[best].d)) break;
		h->c[parent] = h->c[best];
		parent = best;
	}
	h->c[parent] = v;
}
static void fh_emplace(fheap* h, float d, uint32_t id) {
	fh_reserve(h, h->n + 1);
	h->c[h->n].d = d;
	h->c[h->n].id = id;
	h->n++;
	if (h->n >= 2) fh_sift_up(h, h->n - 1);
}
static void fh_pop(fheap* h) { /* :35-37, 96-108 */
	const size_t n = h->n;
	if (n >= 2) {
		const hpair t = h->c[0];
		h->c[0] = h->c[n - 1];
		h->c[n - 1] = t;
		if (n > 2) fh_sift_down(h, 0, n - 1);
	}
	h->n--;
}
static void fh_replace_top(fheap* h, float d, uint32_t id) {
	h->c[0].d = d;
	h->c[0].id = id;
	fh_sift_down(h, 0, h->n);
}

typedef struct {
	int metric;
	size_t n, dim, M, maxM0;
	int maxlevel;
	uint32_t entry;
	size_t num_deleted;
	const uint32_t* links0;
	const uint64_t* upper_off;
	const uint32_t* upper;
	const int32_t* levels;
	const uint64_t* labels;
	const uint8_t* deleted;
	const float* vectors;
	const float* inv_norms;
	/* SQ8 graph (HierarchicalNSWImpl<uint8_t>): codes instead of vectors, the stored corrective offsets, alpha_2; the query travels as codes +
	 * offset (prepareData, hnswalg.h:510-529) and every distance is scaled by normcoef (queryNormCoef, :1855-1863) */
	const uint8_t* codes;
	const float* corr;
	float alpha_2;
	const uint8_t* qcodes;
	float qcorr;
	float normcoef;
} orc_hnsw_graph;

float orc_sq8_dist_query(int metric, size_t dim, float alpha_2, const uint8_t* q, float corr_q, const uint8_t* row, float corr_row,
						 float inv_norm_row);
float orc_sq8_quantize(int metric, size_t dim, float min_q, float alpha, float delta, const float* from, float scale, uint8_t* to);

static float gdist(const orc_hnsw_graph* g, const float* q, uint32_t id) {
	if (g->codes) {
		return g->normcoef * orc_sq8_dist_query(g->metric, g->dim, g->alpha_2, g->qcodes, g->qcorr, g->codes + (size_t)id * g->dim, g->corr[id],
												 g->inv_norms ? g->inv_norms[id] : 1.0f);
	}
	/* normCoef == 1 for the unquantized graph (queryNormCoef, hnswalg.h:1855-1863) */
	return 1.0f * orc_dist(g->metric, q, g->vectors + (size_t)id * g->dim, g->dim, g->inv_norms ? g->inv_norms[id] : 1.0f);
}

static const uint32_t* upper_list(const orc_hnsw_graph* g, uint32_t id, int level) {
	return g->upper + (g->upper_off[id] + (uint64_t)(level - 1)) * (1 + g->M);
}

/* hnswalg.h:799-827 */
static uint32_t entry_point_layer0(const orc_hnsw_graph* g, const float* q, long* ndist) {
	uint32_t cur = g->entry;
	float curdist = gdist(g, q, cur);
	++*ndist;
	for (int level = g->maxlevel; level > 0; level--) {
		int changed = 1;
		while (changed) {
			changed = 0;
			const uint32_t* ll = upper_list(g, cur, level);
			const int size = (int)ll[0];
			for (int i = 0; i < size; i++) {
				const uint32_t cand = ll[1 + i];
				const float d = gdist(g, q, cand);
				++*ndist;
				if (d < curdist) {
					curdist = d;
					cur = cand;
					changed = 1;
				}
			}
		}
	}
	return cur;
}

/* Statistics of the last search on this thread (for the GPU roofline accounting): distance evaluations, hops. */
static __thread long g_last_ndist, g_last_hops;
void orc_hnsw_last_stats(long* ndist, long* hops) {
	*ndist = g_last_ndist;
	*hops = g_last_hops;
}

/* SearchKnn (hnswalg.h:1988-2012). Returns count; out[0] is the best hit (drained like hnsw_index.cc:258-273). */
static size_t search_knn_impl(const orc_hnsw_graph* gp, const float* q, size_t k, size_t ef, float* out_dist, uint64_t* out_label) {
	const orc_hnsw_graph g = *gp;
	const size_t n = g.n, maxM0 = g.maxM0, num_deleted = g.num_deleted;
	const uint32_t* links0 = g.links0;
	const uint64_t* labels = g.labels;
	const uint8_t* deleted = g.deleted;
	if (n == 0) return 0;
	if (k > n) k = n;
	if (!ef) ef = k * 3 / 2;
	long ndist = 0, hops = 0;
	const int bare = num_deleted == 0; /* search(): bare-bone iff no deleted elements (hnswalg.h:1982) */

	const uint32_t ep = entry_point_layer0(&g, q, &ndist);

	/* initLayer0SearchState */
	uint8_t* visited = (uint8_t*)calloc(n, 1);
	fheap top = {0}, cand = {0};
	float lower;
	if (bare || !deleted[ep]) {
		const float d = gdist(&g, q, ep);
		++ndist;
		lower = d;
		fh_emplace(&top, d, ep);
		fh_emplace(&cand, -d, ep);
	} else {
		lower = FLT_MAX;
		fh_emplace(&cand, -lower, ep);
	}
	visited[ep] = 1;

	for (;;) {
		/* layer0ShouldStopBeforePop */
		if (cand.n == 0) break;
		const float cdist = -cand.c[0].d;
		if (bare ? (cdist > lower) : (cdist > lower && top.n >= ef)) break;
		/* runLayer0Step */
		const uint32_t cur = cand.c[0].id;
		fh_pop(&cand);
		const uint32_t* ll = links0 + (size_t)cur * (1 + maxM0);
		const size_t size = ll[0];
		++hops;
		for (size_t j = 0; j < size; j++) {
			const uint32_t cid = ll[1 + j];
			if (visited[cid]) continue;
			visited[cid] = 1;
			const float d = gdist(&g, q, cid);
			++ndist;
			if (top.n < ef || lower > d) {
				fh_emplace(&cand, -d, cid);
				if (bare || !deleted[cid]) {
					if (top.n < ef) {
						fh_emplace(&top, d, cid);
					} else {
						fh_replace_top(&top, d, cid);
					}
				}
				if (top.n) lower = top.c[0].d;
			}
		}
	}
	free(visited);
	free(cand.c);
	g_last_ndist = ndist;
	g_last_hops = hops;

	while (top.n > k) fh_pop(&top);
	/* re-push as (dist, ExternalLabel) into the lexicographic result heap, then drain best-first */
	orc_pair* res = (orc_pair*)malloc((top.n ? top.n : 1) * sizeof(orc_pair));
	size_t rn = 0;
	while (top.n) {
		orc_pair p = {top.c[0].d, labels[top.c[0].id]};
		orc_heap_push(res, &rn, p);
		fh_pop(&top);
	}
	free(top.c);
	const size_t total = rn;
	size_t i = rn;
	while (rn) {
		--i;
		out_dist[i] = res[0].dist;
		out_label[i] = res[0].label;
		orc_heap_pop(res, &rn);
	}
	free(res);
	return total;
}

size_t orc_hnsw_search_knn(int metric, size_t n, size_t dim, size_t M, size_t maxM0, int maxlevel, uint32_t entry, size_t num_deleted,
						   const uint32_t* links0, const uint64_t* upper_off, const uint32_t* upper, const int32_t* levels,
						   const uint64_t* labels, const uint8_t* deleted, const float* vectors, const float* inv_norms, const float* q,
						   size_t k, size_t ef, float* out_dist, uint64_t* out_label) {
	const orc_hnsw_graph g = {metric, n,     dim,      M,     maxM0,  maxlevel, entry,   num_deleted, links0, upper_off, upper,
							  levels, labels, deleted,  vectors, inv_norms, NULL,  NULL,    1.0f,        NULL,   0.0f,      1.0f};
	return search_knn_impl(&g, q, k, ef, out_dist, out_label);
}

/* SearchKnn over the SQ8 graph (HierarchicalNSWImpl<uint8_t>, hnswalg.h:1977-2012): same traversal, distances from the codes.
 * q = the query as the caller hands it to SearchKnn (normalised for cosine), has_qnorm / qnorm = query_data_norm. */
size_t orc_hnsw_search_knn_sq8(int metric, size_t n, size_t dim, size_t M, size_t maxM0, int maxlevel, uint32_t entry, size_t num_deleted,
							   const uint32_t* links0, const uint64_t* upper_off, const uint32_t* upper, const int32_t* levels,
							   const uint64_t* labels, const uint8_t* deleted, const uint8_t* codes, const float* corr, const float* inv_norms,
							   float min_q, float alpha, float alpha_2, float delta, const float* q, int has_qnorm, float qnorm, size_t k, size_t ef,
							   float* out_dist, uint64_t* out_label) {
	if (n == 0) return 0;
	const float normcoef = (metric == ORC_METRIC_COSINE && has_qnorm) ? 1.f / qnorm : 1.f; /* queryNormCoef */
	uint8_t* qcodes = (uint8_t*)malloc(dim ? dim : 1);
	const float qcorr = orc_sq8_quantize(metric, dim, min_q, alpha, delta, q, 1.f / normcoef, qcodes); /* prepareData: norm = 1.f / norm */
	const orc_hnsw_graph g = {metric, n,     dim,     M,    maxM0,     maxlevel, entry, num_deleted, links0, upper_off, upper,
							  levels, labels, deleted, NULL, inv_norms, codes,    corr,  alpha_2,     qcodes, qcorr,     normcoef};
	const size_t r = search_knn_impl(&g, q, k, ef, out_dist, out_label);
	free(qcodes);
	return r;
}

/* ================================================================================================================
 * Streaming (batched) KNN: BeginStreamingSearch / ContinueStreamingSearch (hnswalg.h:1865-1975) with the streaming branches of
 * initLayer0SearchState (:846-850), layer0ShouldStopBeforePop (:865-868) and runLayer0Step (:882-893, 939-940),
 * mergeExtrasIntoTopCandidates (:1893-1926), emitStreamingBatch (:1928-1945).
 * PARITY PINNED: tests/test_oracle_vs_ref.py::test_hnsw_streaming_* replay whole sessions against the real engine (oracle/_ref).
 */
typedef struct {
	orc_hnsw_graph g;
	float* q;
	uint8_t* visited;
	fheap top, extras, cand;
	float lower;
	size_t ef;
	int bare, empty_graph;
} orc_hnsw_stream;

void* orc_hnsw_stream_begin(int metric, size_t n, size_t dim, size_t M, size_t maxM0, int maxlevel, uint32_t entry, size_t num_deleted,
							 const uint32_t* links0, const uint64_t* upper_off, const uint32_t* upper, const int32_t* levels,
							 const uint64_t* labels, const uint8_t* deleted, const float* vectors, const float* inv_norms, const float* q, size_t ef) {
	orc_hnsw_stream* s = (orc_hnsw_stream*)calloc(1, sizeof(*s));
	const orc_hnsw_graph g = {metric, n,     dim,      M,     maxM0,  maxlevel, entry,   num_deleted, links0, upper_off, upper,
							  levels, labels, deleted,  vectors, inv_norms, NULL,  NULL,    1.0f,        NULL,   0.0f,      1.0f};
	s->g = g;
	s->empty_graph = n == 0;
	if (n == 0) return s;
	s->q = (float*)malloc(dim * sizeof(float));
	memcpy(s->q, q, dim * sizeof(float));
	s->ef = ef ? ef : 100; /* kDefaultStreamingEf */
	s->bare = num_deleted == 0;
	long ndist = 0;
	const uint32_t ep = entry_point_layer0(&s->g, s->q, &ndist);
	s->visited = (uint8_t*)calloc(n, 1);
	if (s->bare || !deleted[ep]) {
		const float d = gdist(&s->g, s->q, ep);
		s->lower = d;
		/* streaming: the entry point is NOT put into top_candidates, it gets there when it is popped */
		fh_emplace(&s->cand, -d, ep);
	} else {
		s->lower = FLT_MAX;
		fh_emplace(&s->cand, -s->lower, ep);
	}
	s->visited[ep] = 1;
	return s;
}

static void stream_merge_extras(orc_hnsw_stream* s) { /* hnswalg.h:1893-1926 */
	if (s->top.n >= s->ef || s->extras.n == 0) return;
	if (s->top.n) {
		fheap te = s->extras;
		fheap fresh = {0};
		s->extras = fresh;
		const size_t need = s->ef - s->top.n;
		size_t delta = te.n > need ? te.n - need : 0;
		while (delta-- > 0) {
			fh_emplace(&s->extras, te.c[0].d, te.c[0].id);
			fh_pop(&te);
		}
		while (te.n && s->top.n < s->ef) {
			fh_emplace(&s->top, te.c[0].d, te.c[0].id);
			fh_pop(&te);
		}
		free(te.c);
	} else {
		free(s->top.c);
		s->top = s->extras;
		fheap fresh = {0};
		s->extras = fresh;
		while (s->top.n > s->ef) {
			fh_emplace(&s->extras, s->top.c[0].d, s->top.c[0].id);
			fh_pop(&s->top);
		}
	}
	if (s->top.n) s->lower = s->top.c[0].d;
}

/* Returns the batch size; out_* in emission order (emitStreamingBatch pops the worst first); *exhausted as the reference reports it. */
size_t orc_hnsw_stream_continue(void* h, size_t batch, float* out_dist, uint64_t* out_label, int* exhausted) {
	orc_hnsw_stream* s = (orc_hnsw_stream*)h;
	*exhausted = 0;
	if (batch == 0) return 0;
	if (s->empty_graph) {
		*exhausted = 1;
		return 0;
	}
	const orc_hnsw_graph* g = &s->g;
	const size_t orig_ef = s->ef;
	if (batch > s->ef) s->ef = batch;
	stream_merge_extras(s);
	for (;;) {
		if (s->cand.n == 0) break;
		const float cdist = -s->cand.c[0].d;
		if (cdist > s->lower && s->top.n >= s->ef) break;   /* streaming: never the bare-bone shortcut */
		const float dist = -s->cand.c[0].d;
		const uint32_t cur = s->cand.c[0].id;
		fh_pop(&s->cand);
		if (s->bare || !g->deleted[cur]) {
			if (s->top.n < s->ef) {
				fh_emplace(&s->top, dist, cur);
			} else if (s->lower > dist) {
				const hpair old = s->top.c[0];
				fh_replace_top(&s->top, dist, cur);
				fh_emplace(&s->extras, old.d, old.id);
			}
			s->lower = s->top.c[0].d;
		}
		const uint32_t* ll = g->links0 + (size_t)cur * (1 + g->maxM0);
		const size_t size = ll[0];
		for (size_t j = 0; j < size; j++) {
			const uint32_t cid = ll[1 + j];
			if (s->visited[cid]) continue;
			s->visited[cid] = 1;
			fh_emplace(&s->cand, -gdist(g, s->q, cid), cid);
		}
	}
	s->ef = orig_ef;
	/* emitStreamingBatch */
	fheap tc = s->top;
	fheap fresh = {0};
	s->top = fresh;
	while (tc.n > batch) {
		fh_emplace(&s->top, tc.c[0].d, tc.c[0].id);
		fh_pop(&tc);
	}
	size_t n = 0;
	while (tc.n) {
		out_dist[n] = tc.c[0].d;
		out_label[n] = g->labels[tc.c[0].id];
		++n;
		fh_pop(&tc);
	}
	free(tc.c);
	*exhausted = s->cand.n == 0 && s->top.n == 0 && s->extras.n == 0;
	return n;
}

void orc_hnsw_stream_end(void* h) {
	orc_hnsw_stream* s = (orc_hnsw_stream*)h;
	if (!s) return;
	free(s->q);
	free(s->visited);
	free(s->top.c);
	free(s->extras.c);
	free(s->cand.c);
	free(s);
}
