// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// Flat C wrapper around the REAL IVF backend of the reference: the patched FAISS copy vendored under
// /root/reference/cpp_src/vendor_subdirs/faiss, driven the way IvfIndex drives it (cpp_src/core/index/float_vector/ivf_index.cc):
//   newSpace()                        IndexFlatL2 / IndexFlatIP / IndexFlatCosine                          ivf_index.cc:686-696
//   upsert() -> train + add_with_ids  IndexIVFFlat(space, dim, nCentroids, metric, isCosine), Hashtable map ivf_index.cc:88-108, 469-487
//   search / range_search             faiss::IVFSearchParameters{nprobe}                                    ivf_index.cc:143-272, 355-372
// The FAISS translation units are compiled where they lie (oracle/Makefile: FAISS_TUS); nothing is copied.  FAISS wants BLAS for batched
// distance computations (k-means assignment); the reference loads one at run time through sgemm_dlwrp_ — this file supplies a plain
// triple loop under that name, so training here follows FAISS's algorithm with our summation order inside sgemm, while every per-query
// path (nq = 1 stays below distance_compute_blas_threshold) runs FAISS's own code, which in this copy calls the reference's
// vector_dists::L2SqrDistance / InnerProductDistance (faiss/utils/distances.h:34-42).
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "faiss/IndexFlat.h"
#include "faiss/utils/distances.h"
#include "faiss/IndexIVFFlat.h"
#include "faiss/impl/AuxIndexStructures.h"
#include "faiss/invlists/InvertedLists.h"

#ifndef FINTEGER
#define FINTEGER int
#endif

extern "C" int sgemm_dlwrp_(const char* transa, const char* transb, FINTEGER* m, FINTEGER* n, FINTEGER* k, const float* alpha, const float* a,
							FINTEGER* lda, const float* b, FINTEGER* ldb, float* beta, float* c, FINTEGER* ldc) {
	const bool ta = transa[0] == 'T' || transa[0] == 't', tb = transb[0] == 'T' || transb[0] == 't';
	const int M = *m, N = *n, K = *k;
	for (int j = 0; j < N; ++j) {
		for (int i = 0; i < M; ++i) {
			float s = 0.f;
			for (int l = 0; l < K; ++l) {
				const float av = ta ? a[size_t(i) * *lda + l] : a[size_t(l) * *lda + i];   // column-major storage
				const float bv = tb ? b[size_t(l) * *ldb + j] : b[size_t(j) * *ldb + l];
				s += av * bv;
			}
			float& out = c[size_t(j) * *ldc + i];
			out = *alpha * s + (*beta == 0.f ? 0.f : *beta * out);
		}
	}
	return 0;
}

namespace {
thread_local std::string g_err;
struct IvfRef {
	int metric;   // 0 L2, 1 IP, 2 cosine
	size_t dim, nlist;
	std::unique_ptr<faiss::IndexFlat> space;   // the quantiser of the trained index (ivf_index.cc:97-104 keeps it alive beside map_)
	std::unique_ptr<faiss::IndexIVFFlat> map;
};
std::unique_ptr<faiss::IndexFlat> newSpace(size_t dim, int metric) {
	if (metric == 0) return std::make_unique<faiss::IndexFlatL2>(dim);
	if (metric == 1) return std::make_unique<faiss::IndexFlatIP>(dim);
	return std::make_unique<faiss::IndexFlatCosine>(dim);
}
}  // namespace

extern "C" {

const char* ref_ivf_last_error() { return g_err.c_str(); }

// faiss::distance_compute_blas_threshold (utils/distances.cpp:630): batches of at least this many queries go through sgemm
// (|x|^2 + |y|^2 - 2 x.y); INT_MAX keeps every search — the k-means assignment of training included — on the exact per-pair functions
// (the reference's own L2 / inner-product kernels).  Returns the previous value.
int ref_ivf_set_blas_threshold(int v) {
	const int old = faiss::distance_compute_blas_threshold;
	faiss::distance_compute_blas_threshold = v;
	return old;
}

// The state IvfIndex reaches once more than 39 * nCentroids vectors were upserted: the flat `space_` holding them is trained on and drained
// into the IVF index (ivf_index.cc:96-108).
void* ref_ivf_build(int metric, size_t dim, size_t nlist, size_t n, const float* x, const int64_t* ids) {
	try {
		auto h = std::make_unique<IvfRef>();
		h->metric = metric;
		h->dim = dim;
		h->nlist = nlist;
		auto flat = newSpace(dim, metric);
		flat->add(faiss::idx_t(n), x);
		h->space = newSpace(dim, metric);
		h->map = std::make_unique<faiss::IndexIVFFlat>(h->space.get(), dim, nlist, metric == 0 ? faiss::METRIC_L2 : faiss::METRIC_INNER_PRODUCT,
													   metric == 2);
		h->map->set_direct_map_type(faiss::DirectMap::Type::Hashtable);
		h->map->train(faiss::idx_t(flat->ntotal), flat->get_xb(), flat->get_xb_norms());
		h->map->add_with_ids(faiss::idx_t(flat->ntotal), flat->get_xb(), flat->get_xb_norms(), ids);
		return h.release();
	} catch (const std::exception& e) {
		g_err = e.what();
		return nullptr;
	}
}
void ref_ivf_destroy(void* h) { delete static_cast<IvfRef*>(h); }

// map->search(1, key, k, dists, ids, &params) (ivf_index.cc:152); cosine: the caller passes the normalised key like IvfIndex::select does
int ref_ivf_search(void* h, const float* q, size_t k, size_t nprobe, float* dist, int64_t* labels) {
	try {
		faiss::IVFSearchParameters p;
		p.nprobe = nprobe;
		static_cast<IvfRef*>(h)->map->search(1, q, faiss::idx_t(k), dist, labels, &p);
		return 0;
	} catch (const std::exception& e) {
		g_err = e.what();
		return -1;
	}
}
// map->range_search(1, key, radius, &result, &params) (ivf_index.cc:216): unsorted hits; returns the count (first min(count, cap) written)
long ref_ivf_range(void* h, const float* q, float radius, size_t nprobe, float* dist, int64_t* labels, size_t cap) {
	try {
		faiss::IVFSearchParameters p;
		p.nprobe = nprobe;
		faiss::RangeSearchResult res(1);
		static_cast<IvfRef*>(h)->map->range_search(1, q, radius, &res, &p);
		const size_t n = res.lims[1] - res.lims[0];
		for (size_t i = 0; i < n && i < cap; ++i) {
			dist[i] = res.distances[res.lims[0] + i];
			labels[i] = res.labels[res.lims[0] + i];
		}
		return long(n);
	} catch (const std::exception& e) {
		g_err = e.what();
		return -1;
	}
}
// trained state: centroids [nlist][dim], list sizes [nlist]
void ref_ivf_export(void* h, float* centroids, uint64_t* listSizes) {
	auto* r = static_cast<IvfRef*>(h);
	std::memcpy(centroids, r->space->get_xb(), r->nlist * r->dim * sizeof(float));
	for (size_t l = 0; l < r->nlist; ++l) listSizes[l] = r->map->invlists->list_size(l);
}
// ids of one inverted list, in list order
void ref_ivf_list_ids(void* h, size_t list, int64_t* out) {
	auto* r = static_cast<IvfRef*>(h);
	const size_t n = r->map->invlists->list_size(list);
	faiss::InvertedLists::ScopedIds ids(r->map->invlists, list);
	for (size_t i = 0; i < n; ++i) out[i] = ids[i];
}
long ref_ivf_remove(void* h, const int64_t* ids, size_t n) {
	try {
		size_t removed = 0;
		for (size_t i = 0; i < n; ++i) removed += static_cast<IvfRef*>(h)->map->remove_ids(faiss::IDSelectorArray{1, &ids[i]});   // ivf_index.cc:124
		return long(removed);
	} catch (const std::exception& e) {
		g_err = e.what();
		return -1;
	}
}

}  // extern "C"
