// C-ABI of the BM25 merge (include/rxgpu.h, rxgpu_ft_*): device mirror of the ft_fast posting lists + the scoring launch.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/rxgpu.h"
#include "rxgpu_internal.h"

using rxgpu::set_error;

#define RX_HIP(expr)                                                                \
	do {                                                                            \
		hipError_t e__ = (expr);                                                    \
		if (e__ != hipSuccess) {                                                    \
			set_error(std::string(#expr) + ": " + hipGetErrorString(e__));          \
			return e__ == hipErrorOutOfMemory ? RXGPU_ERR_NOMEM : RXGPU_ERR_DEVICE; \
		}                                                                           \
	} while (0)
#define RX_CHECK(cond, code, msg) \
	do {                          \
		if (!(cond)) {            \
			set_error(msg);       \
			return code;          \
		}                         \
	} while (0)

struct rxgpu_ft_word {
	uint64_t n = 0, nent = 0;
	uint32_t* doc = nullptr;
	uint32_t* ent_off = nullptr;
	uint8_t* ent_field = nullptr;
	uint32_t* ent_tf = nullptr;
	uint32_t* ent_first_pos = nullptr;
	void release() {
		for (void* p : {static_cast<void*>(doc), static_cast<void*>(ent_off), static_cast<void*>(ent_field), static_cast<void*>(ent_tf),
						static_cast<void*>(ent_first_pos)}) {
			if (p) (void)hipFree(p);
		}
		*this = rxgpu_ft_word{};
	}
};

struct rxgpu_ft_index {
	int device = 0;
	uint32_t num_fields = 0;
	uint64_t total_docs = 0;
	float* d_words = nullptr;
	float* d_avg = nullptr;
	uint8_t* d_removed = nullptr;
	std::unordered_map<uint32_t, rxgpu_ft_word> words;
	std::mutex mtx;
	hipStream_t stream = nullptr;
	rxgpu_devbuf d_best, d_first, d_pfield, d_blocks, d_total, d_out_doc, d_out_proc, d_out_field, d_excluded, d_cfg, d_subs;
	uint64_t stat_postings = 0;
	double stat_ms = 0.0;
};

namespace {
struct DevGuard {
	int prev = -1;
	explicit DevGuard(int dev) {
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		if (prev != dev) (void)hipSetDevice(dev);
	}
	~DevGuard() {
		if (prev >= 0) (void)hipSetDevice(prev);
	}
};
template <typename T>
int upload(T*& dst, const T* src, size_t count) {
	if (dst) (void)hipFree(dst);
	dst = nullptr;
	if (!count) return RXGPU_OK;
	RX_HIP(hipMalloc(reinterpret_cast<void**>(&dst), count * sizeof(T)));
	RX_HIP(hipMemcpy(dst, src, count * sizeof(T), hipMemcpyHostToDevice));
	return RXGPU_OK;
}
}  // namespace

extern "C" {

int rxgpu_ft_create(uint32_t num_fields, int device, rxgpu_ft_index** out) {
	RX_CHECK(out, RXGPU_ERR_PARAMS, "rxgpu_ft_create: out is null");
	RX_CHECK(num_fields >= 1 && num_fields <= 63, RXGPU_ERR_PARAMS, "rxgpu_ft_create: 1..63 fields (kMaxFtCompositeFields)");
	int ndev = 0;
	RX_HIP(hipGetDeviceCount(&ndev));
	RX_CHECK(device >= 0 && device < ndev, RXGPU_ERR_PARAMS, "rxgpu_ft_create: no such device");
	DevGuard dg(device);
	auto* h = new rxgpu_ft_index();
	h->device = device;
	h->num_fields = num_fields;
	if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
		delete h;
		set_error("hipStreamCreateWithFlags failed");
		return RXGPU_ERR_DEVICE;
	}
	*out = h;
	return RXGPU_OK;
}

void rxgpu_ft_destroy(rxgpu_ft_index* h) {
	if (!h) return;
	DevGuard dg(h->device);
	(void)hipDeviceSynchronize();
	for (auto& kv : h->words) kv.second.release();
	for (void* p : {static_cast<void*>(h->d_words), static_cast<void*>(h->d_avg), static_cast<void*>(h->d_removed)}) {
		if (p) (void)hipFree(p);
	}
	for (rxgpu_devbuf* b : {&h->d_best, &h->d_first, &h->d_pfield, &h->d_blocks, &h->d_total, &h->d_out_doc, &h->d_out_proc, &h->d_out_field,
							&h->d_excluded, &h->d_cfg, &h->d_subs}) {
		b->release();
	}
	if (h->stream) (void)hipStreamDestroy(h->stream);
	delete h;
}

int rxgpu_ft_set_docs(rxgpu_ft_index* h, uint64_t total_docs, const float* words_in_field, const float* avg_words, const uint8_t* removed) {
	RX_CHECK(h && words_in_field && avg_words, RXGPU_ERR_PARAMS, "rxgpu_ft_set_docs: null argument");
	RX_CHECK(total_docs >= 1 && total_docs < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "rxgpu_ft_set_docs: total_docs out of range");
	std::lock_guard<std::mutex> lk(h->mtx);
	DevGuard dg(h->device);
	RX_HIP(hipStreamSynchronize(h->stream));
	if (int rc = upload(h->d_words, words_in_field, total_docs * h->num_fields); rc) return rc;
	if (int rc = upload(h->d_avg, avg_words, h->num_fields); rc) return rc;
	if (removed) {
		if (int rc = upload(h->d_removed, removed, total_docs); rc) return rc;
	} else {
		if (h->d_removed) (void)hipFree(h->d_removed);
		h->d_removed = nullptr;
	}
	h->total_docs = total_docs;
	return RXGPU_OK;
}

int rxgpu_ft_set_word(rxgpu_ft_index* h, uint32_t word_id, uint64_t n, const uint32_t* doc, const uint32_t* ent_off, const uint8_t* ent_field,
					  const uint32_t* ent_tf, const uint32_t* ent_first_pos) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null ft index");
	RX_CHECK(n == 0 || (doc && ent_off && ent_field && ent_tf && ent_first_pos), RXGPU_ERR_PARAMS, "rxgpu_ft_set_word: null argument");
	std::lock_guard<std::mutex> lk(h->mtx);
	DevGuard dg(h->device);
	RX_HIP(hipStreamSynchronize(h->stream));
	rxgpu_ft_word& w = h->words[word_id];
	w.release();
	if (n == 0) return RXGPU_OK;
	const uint64_t nent = ent_off[n];
	if (int rc = upload(w.doc, doc, n); rc) return rc;
	if (int rc = upload(w.ent_off, ent_off, n + 1); rc) return rc;
	if (int rc = upload(w.ent_field, ent_field, nent); rc) return rc;
	if (int rc = upload(w.ent_tf, ent_tf, nent); rc) return rc;
	if (int rc = upload(w.ent_first_pos, ent_first_pos, nent); rc) return rc;
	w.n = n;
	w.nent = nent;
	return RXGPU_OK;
}

int rxgpu_ft_merge_simple_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_term_opts* opts, uint32_t nsub,
							  const uint32_t* word_ids, const float* procs, const uint8_t* excluded, uint32_t* out_doc, float* out_proc,
							  uint8_t* out_field, uint64_t cap, uint64_t* out_n) {
	RX_CHECK(h && cfg && opts && out_n, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: null argument");
	*out_n = 0;
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, "rxgpu_ft_merge_simple_raw: rxgpu_ft_set_docs was not called");
	if (nsub == 0) return RXGPU_OK;
	RX_CHECK(word_ids && procs, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: null argument");
	std::lock_guard<std::mutex> lk(h->mtx);
	DevGuard dg(h->device);
	std::vector<rxgpu::FtSubterm> subs(nsub);
	uint64_t total = 0;
	uint32_t nblocks = 0;
	for (uint32_t s = 0; s < nsub; ++s) {
		auto it = h->words.find(word_ids[s]);
		RX_CHECK(it != h->words.end(), RXGPU_ERR_NOTFOUND, "rxgpu_ft_merge_simple_raw: unknown word id");
		const rxgpu_ft_word& w = it->second;
		rxgpu::FtSubterm& ft = subs[s];
		ft.n = w.n;
		ft.doc = w.doc;
		ft.ent_off = w.ent_off;
		ft.ent_field = w.ent_field;
		ft.ent_tf = w.ent_tf;
		ft.ent_first_pos = w.ent_first_pos;
		// Bm25Rx::IDF(totalDocCount = totalNumDocs - 1, matchedDocCount = |postings|)  (bm25.h:19-26, mergerimpl.h:203-205)
		const double td = double(h->total_docs - 1), md = double(w.n);
		double f = w.n ? std::log((td - md + 1) / md) / std::log(1 + td) : 0.2;
		if (f < 0.2) f = 0.2;
		ft.idf = f;
		ft.proc = procs[s];
		ft.gp_base = total;
		ft.block_base = nblocks;
		total += w.n;
		nblocks += rxgpu::bm25_scan_blocks_for(w.n);
	}
	RX_CHECK(total < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: more than 2^32 postings in one merge");
	const uint64_t max_merged = std::min<uint64_t>(cfg->merge_limit, total);   // Merge(): min(mergeLimit, totalORVids)
	if (max_merged == 0) return RXGPU_OK;
	RX_CHECK(cap >= max_merged && out_doc && out_proc && out_field, RXGPU_ERR_OVERFLOW, "rxgpu_ft_merge_simple_raw: output buffers too small");

	const uint32_t nf = h->num_fields;
	// per-field parameters as floats (bound() takes float arguments), packed in one upload
	std::vector<float> fcfg(size_t(7) * nf);
	for (uint32_t f = 0; f < nf; ++f) {
		fcfg[0 * nf + f] = opts->field_boost[f];
		fcfg[1 * nf + f] = float(cfg->bm25_boost[f]);
		fcfg[2 * nf + f] = float(cfg->bm25_weight[f]);
		fcfg[3 * nf + f] = float(cfg->term_len_boost[f]);
		fcfg[4 * nf + f] = float(cfg->term_len_weight[f]);
		fcfg[5 * nf + f] = float(cfg->position_boost[f]);
		fcfg[6 * nf + f] = float(cfg->position_weight[f]);
	}
	const size_t cfg_bytes = fcfg.size() * sizeof(float) + nf;
	if (int rc = h->d_cfg.ensure(cfg_bytes); rc) return rc;
	if (int rc = h->d_best.ensure(h->total_docs * sizeof(unsigned long long)); rc) return rc;
	if (int rc = h->d_first.ensure(h->total_docs * sizeof(uint32_t)); rc) return rc;
	if (int rc = h->d_pfield.ensure(total); rc) return rc;
	if (int rc = h->d_blocks.ensure(size_t(nblocks) * sizeof(uint32_t)); rc) return rc;
	if (int rc = h->d_total.ensure(sizeof(uint32_t)); rc) return rc;
	if (int rc = h->d_out_doc.ensure(max_merged * sizeof(uint32_t)); rc) return rc;
	if (int rc = h->d_out_proc.ensure(max_merged * sizeof(float)); rc) return rc;
	if (int rc = h->d_out_field.ensure(max_merged); rc) return rc;
	hipStream_t st = h->stream;
	RX_HIP(hipMemcpyAsync(h->d_cfg.ptr, fcfg.data(), fcfg.size() * sizeof(float), hipMemcpyHostToDevice, st));
	RX_HIP(hipMemcpyAsync(static_cast<char*>(h->d_cfg.ptr) + fcfg.size() * sizeof(float), opts->need_sum_rank, nf, hipMemcpyHostToDevice, st));
	const uint8_t* d_excl = nullptr;
	if (excluded) {
		if (int rc = h->d_excluded.ensure(h->total_docs); rc) return rc;
		RX_HIP(hipMemcpyAsync(h->d_excluded.ptr, excluded, h->total_docs, hipMemcpyHostToDevice, st));
		d_excl = static_cast<const uint8_t*>(h->d_excluded.ptr);
	}
	RX_HIP(hipMemsetAsync(h->d_best.ptr, 0, h->total_docs * sizeof(unsigned long long), st));
	RX_HIP(hipMemsetAsync(h->d_first.ptr, 0xFF, h->total_docs * sizeof(uint32_t), st));

	rxgpu::FtMergeParams p{};
	const float* fc = static_cast<const float*>(h->d_cfg.ptr);
	p.num_fields = nf;
	p.words = h->d_words;
	p.avg_words = h->d_avg;
	p.removed = h->d_removed;
	p.excluded = d_excl;
	p.k1 = cfg->bm25_k1;
	p.b = cfg->bm25_b;
	p.summation_ratio = cfg->summation_ranks_by_fields_ratio;
	p.opts_boost = opts->boost;
	p.term_len_boost_in = opts->term_len_boost;
	p.field_boost = fc + 0 * nf;
	p.bm25_boost = fc + 1 * nf;
	p.bm25_weight = fc + 2 * nf;
	p.term_len_boost = fc + 3 * nf;
	p.term_len_weight = fc + 4 * nf;
	p.position_boost = fc + 5 * nf;
	p.position_weight = fc + 6 * nf;
	p.need_sum_rank = reinterpret_cast<const uint8_t*>(fc + 7 * nf);
	p.best = static_cast<unsigned long long*>(h->d_best.ptr);
	p.first = static_cast<uint32_t*>(h->d_first.ptr);
	p.pfield = static_cast<uint8_t*>(h->d_pfield.ptr);
	p.max_merged = uint32_t(max_merged);
	p.out_doc = static_cast<uint32_t*>(h->d_out_doc.ptr);
	p.out_proc = static_cast<float*>(h->d_out_proc.ptr);
	p.out_field = static_cast<uint8_t*>(h->d_out_field.ptr);

	hipEvent_t e0, e1;
	RX_HIP(hipEventCreate(&e0));
	RX_HIP(hipEventCreate(&e1));
	RX_HIP(hipEventRecord(e0, st));
	if (int rc = h->d_subs.ensure(subs.size() * sizeof(rxgpu::FtSubterm)); rc) return rc;
	RX_HIP(hipMemcpyAsync(h->d_subs.ptr, subs.data(), subs.size() * sizeof(rxgpu::FtSubterm), hipMemcpyHostToDevice, st));
	RX_HIP(hipEventRecord(e0, st));   // re-record: time only the scoring launch
	rxgpu::launch_bm25_score_fused(p, static_cast<const rxgpu::FtSubterm*>(h->d_subs.ptr), nsub, total, st);
	RX_HIP(hipEventRecord(e1, st));
	for (const auto& s : subs) rxgpu::launch_bm25_count_adds(p, s, static_cast<uint32_t*>(h->d_blocks.ptr), st);
	rxgpu::launch_bm25_scan_blocks(static_cast<uint32_t*>(h->d_blocks.ptr), nblocks, static_cast<uint32_t*>(h->d_total.ptr), st);
	for (const auto& s : subs) rxgpu::launch_bm25_emit(p, s, static_cast<const uint32_t*>(h->d_blocks.ptr), st);
	RX_HIP(hipGetLastError());
	uint32_t distinct = 0;
	RX_HIP(hipMemcpyAsync(&distinct, h->d_total.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	RX_HIP(hipStreamSynchronize(st));
	float ms = 0.f;
	(void)hipEventElapsedTime(&ms, e0, e1);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	h->stat_postings += total;
	h->stat_ms += ms;
	const uint64_t n = std::min<uint64_t>(distinct, max_merged);
	if (n) {
		RX_HIP(hipMemcpy(out_doc, p.out_doc, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
		RX_HIP(hipMemcpy(out_proc, p.out_proc, n * sizeof(float), hipMemcpyDeviceToHost));
		RX_HIP(hipMemcpy(out_field, p.out_field, n, hipMemcpyDeviceToHost));
	}
	*out_n = n;
	return RXGPU_OK;
}

int rxgpu_ft_read_stats(rxgpu_ft_index* h, uint64_t* postings, double* kernel_ms) {
	RX_CHECK(h && postings && kernel_ms, RXGPU_ERR_PARAMS, "rxgpu_ft_read_stats: null argument");
	std::lock_guard<std::mutex> lk(h->mtx);
	*postings = h->stat_postings;
	*kernel_ms = h->stat_ms;
	h->stat_postings = 0;
	h->stat_ms = 0.0;
	return RXGPU_OK;
}

}  // extern "C"
