/* TEST INFRASTRUCTURE ONLY — this file is the CHECKER, never the product.
 *
 * Plain-C restatement of the reference's float_vector brute-force KNN path (Restream/reindexer v5.15.0).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 *
 * PARITY PINNED: every function here is checked bit-for-bit against the real reference engines
 * (oracle/_ref/libref_oracle.so, built in place from /root/reference by `make -C oracle ref`) in
 * tests/test_oracle_vs_ref.py, and against the committed fixtures under tests/golden/ generated from
 * that same build by tests/golden/make_golden.py.  The reference's own tests hold no golden vectors for
 * KNN (random unseeded data, gtests/tools.h:121-129), so the reference binary itself is the anchor.
 *
 * Arithmetic model (reference built with g++ 11.4 -O2, RX_TARGET_INSTRUCTIONS=avx512):
 *   cpp_src/tools/distances/l2_dist.cc:38-72   L2SqrAVX512
 *   cpp_src/tools/distances/ip_dist.cc:31-70   InnerProductAVX512
 * 4 zmm accumulators x 16 lanes = 64 independent fmaf chains, chain L owning elements i == L (mod 64);
 * lane-wise (s0+s1)+(s2+s3); IP only: a 16-wide fmaf loop for the [64*floor(D/64), 16*floor(D/16)) span;
 * _mm512_reduce_add_ps = fold 16->8->4, then (t0+t2)+(t1+t3); finally "+ scalar tail" where the scalar tail
 * (l2_dist.cc:12-26 / ip_dist.cc:10-20, compiled under the FAISS imprecise pragma) is, in this build, a
 * sequential fmaf chain starting from 0.  The tail and normalisation orders are compiler-defined in the
 * reference; they are properties of the pinned oracle build and re-asserted by the tests above.
 */
#include "oracle_knn.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__) && defined(__GNUC__)
#define ORC_HOT __attribute__((target_clones("arch=haswell", "default")))
#else
#define ORC_HOT
#endif

/* _mm512_reduce_add_ps as emitted by GCC 11 (avx512fintrin.h:16112-16128) */
static inline float reduce16(const float* v) {
	float t8[8], t4[4];
	for (int j = 0; j < 8; ++j) t8[j] = v[j] + v[j + 8];
	for (int j = 0; j < 4; ++j) t4[j] = t8[j] + t8[j + 4];
	return (t4[0] + t4[2]) + (t4[1] + t4[3]);
}

/* cpp_src/tools/distances/l2_dist.cc:38-72 (+ scalar tail :12-26) */
ORC_HOT float orc_l2sqr(const float* a, const float* b, size_t d) {
	float s[64];
	float v[16];
	const size_t simd_end = d & ~(size_t)63;
	memset(s, 0, sizeof(s));
	for (size_t i = 0; i < simd_end; i += 64) {
		for (int l = 0; l < 64; ++l) {
			const float df = a[i + l] - b[i + l];
			s[l] = fmaf(df, df, s[l]);
		}
	}
	for (int j = 0; j < 16; ++j) v[j] = (s[j] + s[16 + j]) + (s[32 + j] + s[48 + j]);
	float tail = 0.0f;
	for (size_t i = simd_end; i < d; ++i) {
		const float df = a[i] - b[i];
		tail = fmaf(df, df, tail);
	}
	return reduce16(v) + tail;
}

/* cpp_src/tools/distances/ip_dist.cc:31-70 (+ scalar tail :10-20) */
ORC_HOT float orc_ip(const float* a, const float* b, size_t d) {
	float s[64];
	float v[16];
	const size_t simd_end = d & ~(size_t)15;
	size_t i = 0;
	memset(s, 0, sizeof(s));
	for (; i + 64 <= simd_end; i += 64) {
		for (int l = 0; l < 64; ++l) s[l] = fmaf(a[i + l], b[i + l], s[l]);
	}
	for (int j = 0; j < 16; ++j) v[j] = (s[j] + s[16 + j]) + (s[32 + j] + s[48 + j]);
	for (; i < simd_end; i += 16) {
		for (int j = 0; j < 16; ++j) v[j] = fmaf(a[i + j], b[i + j], v[j]);
	}
	float tail = 0.0f;
	for (i = simd_end; i < d; ++i) tail = fmaf(a[i], b[i], tail);
	return reduce16(v) + tail;
}

void orc_l2sqr_many(const float* q, const float* rows, size_t n, size_t d, float* out) {
	for (size_t i = 0; i < n; ++i) out[i] = orc_l2sqr(q, rows + i * d, d);
}
void orc_ip_many(const float* q, const float* rows, size_t n, size_t d, float* out) {
	for (size_t i = 0; i < n; ++i) out[i] = orc_ip(q, rows + i * d, d);
}

/* cpp_src/tools/normalize.cc:10-23 calculateL2Module: k = 1/sqrt(sum x^2) unless the vector is already unit
 * (|1 - sum| <= 1e-5) or zero; "1.0 / std::sqrt(float)" is a double division rounded to float. */
ORC_HOT float orc_l2_module(const float* x, int32_t d) {
	float sq = 0.0f;
	for (int32_t i = 0; i < d; ++i) sq = fmaf(x[i], x[i], sq);
	float k = 1.0f;
	if (sq > 0.0f && fabsf(1.0f - sq) > 0.00001f) {
		k = (float)(1.0 / (double)sqrtf(sq));
	}
	return k;
}

/* cpp_src/tools/normalize.h:18-22 NormalizeCopyVector + normalize.cc:25-32 */
float orc_normalize_copy(const float* x, int32_t d, float* out) {
	const float k = orc_l2_module(x, d);
	for (int32_t i = 0; i < d; ++i) out[i] = x[i] * k;
	return k;
}

/* cpp_src/core/index/float_vector/hnswlib/hnswlib.h:147-165,192-197 DistCalculator<float>::operator()(q,row,id):
 * alpha2 == 1, corrective offsets == 0 for fp32; IP/cosine are negated; cosine multiplies by the stored 1/|row|. */
float orc_dist(int metric, const float* q, const float* row, size_t d, float inv_norm) {
	if (metric == ORC_METRIC_L2) {
		return 1.0f * orc_l2sqr(q, row, d) + 0.0f + 0.0f;
	}
	float dist = -(1.0f * orc_ip(q, row, d) + 0.0f + 0.0f);
	if (metric == ORC_METRIC_COSINE) dist *= inv_norm;
	return dist;
}

/* ---- cpp_src/core/index/float_vector/hnswlib/priority_queue.h:7-152, comparator std::less<pair<float,u64>> ---- */
static inline int pair_less(orc_pair a, orc_pair b) { return a.dist < b.dist || (!(b.dist < a.dist) && a.label < b.label); }

static void sift_up(orc_pair* c, size_t child) { /* priority_queue.h:109-123 */
	const orc_pair value = c[child];
	while (child > 0) {
		const size_t parent = (child - 1) / 2;
		if (!pair_less(c[parent], value)) break;
		c[child] = c[parent];
		child = parent;
	}
	c[child] = value;
}
static void sift_down(orc_pair* c, size_t parent, size_t heap_size) { /* priority_queue.h:125-151 */
	const orc_pair value = c[parent];
	for (;;) {
		const size_t left = parent * 2 + 1;
		if (left >= heap_size) break;
		size_t best = left;
		const size_t right = left + 1;
		if (right < heap_size && pair_less(c[left], c[right])) best = right;
		if (!pair_less(value, c[best])) break;
		c[parent] = c[best];
		parent = best;
	}
	c[parent] = value;
}
void orc_heap_push(orc_pair* c, size_t* n, orc_pair v) { /* :18-33,87-94 */
	c[(*n)++] = v;
	if (*n >= 2) sift_up(c, *n - 1);
}
void orc_heap_pop(orc_pair* c, size_t* n) { /* :35-37,96-108 */
	const size_t sz = *n;
	if (sz >= 2) {
		const orc_pair t = c[0];
		c[0] = c[sz - 1];
		c[sz - 1] = t;
		if (sz > 2) sift_down(c, 0, sz - 1);
	}
	--*n;
}
void orc_heap_replace_top(orc_pair* c, size_t n, orc_pair v) { /* :39-57 */
	c[0] = v;
	sift_down(c, 0, n);
}

/* drain back-to-front => out[0] = best (hnsw_index.cc:258-273 pops the heap into index i = size-1 .. 0) */
static size_t drain(orc_pair* heap, size_t n, float* out_dist, uint64_t* out_label, size_t cap) {
	const size_t total = n;
	size_t i = n;
	while (n) {
		--i;
		if (i < cap) {
			out_dist[i] = heap[0].dist;
			out_label[i] = heap[0].label;
		}
		orc_heap_pop(heap, &n);
	}
	return total;
}

/* cpp_src/core/index/float_vector/hnswlib/bruteforce.cc:103-127 BruteforceSearch::SearchKnn.
 * rows: [n][d] in INTERNAL index order (the reference's AoS row is [d floats][u64 label], bruteforce.h:47-48);
 * inv_norms: per-row 1/|row| (cosine only).  Returns count; out[0] is the best hit. */
size_t orc_bf_search_knn(int metric, const float* rows, const uint64_t* labels, const float* inv_norms, size_t n, size_t d,
						 const float* q, size_t k, float* out_dist, uint64_t* out_label) {
	if (n == 0 || k == 0) return 0;
	if (k > n) k = n;
	orc_pair* heap = (orc_pair*)malloc(k * sizeof(orc_pair));
	size_t hn = 0;
	for (size_t i = 0; i < k; ++i) {
		orc_pair p = {orc_dist(metric, q, rows + i * d, d, inv_norms ? inv_norms[i] : 1.0f), labels[i]};
		orc_heap_push(heap, &hn, p);
	}
	float lastdist = heap[0].dist;
	for (size_t i = k; i < n; ++i) {
		const float dist = orc_dist(metric, q, rows + i * d, d, inv_norms ? inv_norms[i] : 1.0f);
		if (dist < lastdist) {
			orc_pair p = {dist, labels[i]};
			orc_heap_replace_top(heap, hn, p);
			lastdist = heap[0].dist;
		}
	}
	const size_t cnt = drain(heap, hn, out_dist, out_label, k);
	free(heap);
	return cnt;
}

/* cpp_src/core/index/float_vector/hnswlib/bruteforce.cc:129-143 BruteforceSearch::SearchRange
 * (radius already negated by the caller for IP/cosine, hnsw_index.cc:185).  Returns the TOTAL number of
 * hits; at most cap are written (best first). */
size_t orc_bf_search_range(int metric, const float* rows, const uint64_t* labels, const float* inv_norms, size_t n, size_t d,
						   const float* q, float radius, float* out_dist, uint64_t* out_label, size_t cap) {
	size_t hcap = 64, hn = 0;
	orc_pair* heap = (orc_pair*)malloc(hcap * sizeof(orc_pair));
	for (size_t i = 0; i < n; ++i) {
		const float dist = orc_dist(metric, q, rows + i * d, d, inv_norms ? inv_norms[i] : 1.0f);
		if (dist < radius) {
			if (hn == hcap) {
				hcap *= 2;
				heap = (orc_pair*)realloc(heap, hcap * sizeof(orc_pair));
			}
			orc_pair p = {dist, labels[i]};
			orc_heap_push(heap, &hn, p);
		}
	}
	const size_t cnt = drain(heap, hn, out_dist, out_label, cap);
	free(heap);
	return cnt;
}

typedef struct {
	int metric;
	const float* rows;
	const uint64_t* labels;
	const float* inv_norms;
	size_t n, d;
	const float* queries;
	size_t nq, k;
	float* out_dist;
	uint64_t* out_label;
	size_t* out_count;
	int tid, threads;
} batch_arg;

static void* batch_worker(void* p) {
	batch_arg* a = (batch_arg*)p;
	for (size_t qi = (size_t)a->tid; qi < a->nq; qi += (size_t)a->threads) {
		const size_t c = orc_bf_search_knn(a->metric, a->rows, a->labels, a->inv_norms, a->n, a->d, a->queries + qi * a->d, a->k,
										   a->out_dist + qi * a->k, a->out_label + qi * a->k);
		if (a->out_count) a->out_count[qi] = c;
	}
	return NULL;
}

/* T concurrent query threads over one shared index — the reference's own concurrency model (it has no
 * intra-query parallelism; cf. gtests/tests/unit/float_vector_index.cc:258-294 runMultithreadQueries). */
void orc_bf_search_knn_batch(int metric, const float* rows, const uint64_t* labels, const float* inv_norms, size_t n, size_t d,
							 const float* queries, size_t nq, size_t k, float* out_dist, uint64_t* out_label, size_t* out_count,
							 int threads) {
	if (threads < 1) threads = 1;
	pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
	batch_arg* args = (batch_arg*)malloc((size_t)threads * sizeof(batch_arg));
	for (int t = 0; t < threads; ++t) {
		batch_arg a = {metric, rows, labels, inv_norms, n, d, queries, nq, k, out_dist, out_label, out_count, t, threads};
		args[t] = a;
		if (t > 0) pthread_create(&th[t], NULL, batch_worker, &args[t]);
	}
	batch_worker(&args[0]);
	for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
	free(args);
	free(th);
}

static int cmp_i32(const void* a, const void* b) {
	const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
	return (x > y) - (x < y);
}

/* cpp_src/core/index/float_vector/hnsw_index.cc:231-288 HnswIndexBase<Map>::select post-processing, applied to
 * an engine result already drained best-first (dist[0] best):
 *   rank = dist (L2) or -dist (IP/cosine) (:261-270); rowId = label >> 32 (:271);
 *   NeedSort: ids inside runs of exactly equal rank are sorted ascending (:239-257, 274-276);
 *   IsArray: duplicates of a rowId are dropped keeping the first = best (float_vector_index.h:140-160);
 *   k AND radius both given: truncate to k (removeOverK, :193-203).
 * Returns the number of (id, rank) pairs written.
 * PARITY PINNED: tests/test_select_pin.py runs it against the reference's HnswIndexBase<BruteforceSearch>::select compiled in place
 * (oracle/_ref/libref_select.so, oracle/ref/ref_select_shim.cc). */
size_t orc_select_postprocess(int metric, const float* dist, const uint64_t* label, size_t n, int need_sort, int is_array,
							  int has_k, size_t k, int has_radius, int32_t* out_ids, float* out_ranks) {
	if (n == 0) return 0;
	for (size_t i = 0; i < n; ++i) {
		out_ranks[i] = metric == ORC_METRIC_L2 ? dist[i] : -dist[i];
		out_ids[i] = (int32_t)(label[i] >> 32);
	}
	if (need_sort) {
		/* the reference walks i = n-1 .. 0 and sorts (i, lastSameDist] whenever rank[i] is strictly better
		 * than rank[lastSameDist]; the net effect is: every maximal run of equal ranks is sorted by id. */
		size_t last_same = n - 1;
		for (size_t ii = n; ii-- > 0;) {
			const int new_dist = metric == ORC_METRIC_L2 ? (out_ranks[last_same] > out_ranks[ii]) : (out_ranks[last_same] < out_ranks[ii]);
			if (new_dist) {
				qsort(out_ids + ii + 1, last_same - ii, sizeof(int32_t), cmp_i32);
				last_same = ii;
			}
		}
		qsort(out_ids, last_same + 1, sizeof(int32_t), cmp_i32);
	}
	size_t cnt = n;
	if (is_array) {
		/* removeDuplicateRowId: keep the first occurrence of each rowId, preserve order */
		size_t w = 0;
		for (size_t i = 0; i < n; ++i) {
			int dup = 0;
			for (size_t j = 0; j < w; ++j) {
				if (out_ids[j] == out_ids[i]) {
					dup = 1;
					break;
				}
			}
			if (!dup) {
				out_ids[w] = out_ids[i];
				out_ranks[w] = out_ranks[i];
				++w;
			}
		}
		cnt = w;
	}
	if (has_k && has_radius && cnt > k) cnt = k;
	return cnt;
}
