/* TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's float_vector KNN path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * See oracle_knn.c for the reference file:line each function follows. */
#ifndef ORACLE_KNN_H
#define ORACLE_KNN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_METRIC_L2 = 0, ORC_METRIC_IP = 1, ORC_METRIC_COSINE = 2 };

float orc_l2sqr(const float* a, const float* b, size_t d);
float orc_ip(const float* a, const float* b, size_t d);
void orc_l2sqr_many(const float* q, const float* rows, size_t n, size_t d, float* out);
void orc_ip_many(const float* q, const float* rows, size_t n, size_t d, float* out);
float orc_l2_module(const float* x, int32_t d);
float orc_normalize_copy(const float* x, int32_t d, float* out);
float orc_dist(int metric, const float* q, const float* row, size_t d, float inv_norm);

size_t orc_bf_search_knn(int metric, const float* rows, const uint64_t* labels, const float* inv_norms, size_t n, size_t d,
						 const float* q, size_t k, float* out_dist, uint64_t* out_label);
size_t orc_bf_search_range(int metric, const float* rows, const uint64_t* labels, const float* inv_norms, size_t n, size_t d,
						   const float* q, float radius, float* out_dist, uint64_t* out_label, size_t cap);
/* multi-threaded driver over independent queries (the reference's own concurrency model) */
void orc_bf_search_knn_batch(int metric, const float* rows, const uint64_t* labels, const float* inv_norms, size_t n, size_t d,
							 const float* queries, size_t nq, size_t k, float* out_dist, uint64_t* out_label, size_t* out_count,
							 int threads);

size_t orc_select_postprocess(int metric, const float* dist, const uint64_t* label, size_t n, int need_sort, int is_array,
							  int has_k, size_t k, int has_radius, int32_t* out_ids, float* out_ranks);

/* (dist,label) max-heap with the reference's exact sift mechanics, exposed for tests */
typedef struct {
	float dist;
	uint64_t label;
} orc_pair;
void orc_heap_push(orc_pair* c, size_t* n, orc_pair v);
void orc_heap_pop(orc_pair* c, size_t* n);
void orc_heap_replace_top(orc_pair* c, size_t n, orc_pair v);

#ifdef __cplusplus
}
#endif
#endif
