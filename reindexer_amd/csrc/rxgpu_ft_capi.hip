// C-ABI of the BM25 merge (include/rxgpu.h, rxgpu_ft_*): device mirror of the ft_fast posting lists + the scoring launch.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdint>
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
	uint32_t* pos_off = nullptr;   // only for words uploaded with their positions (multi-term merge)
	uint64_t* fpos = nullptr;
	void release() {
		for (void* p : {static_cast<void*>(doc), static_cast<void*>(ent_off), static_cast<void*>(ent_field), static_cast<void*>(ent_tf),
						static_cast<void*>(ent_first_pos), static_cast<void*>(pos_off), static_cast<void*>(fpos)}) {
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
	rxgpu_devbuf d_excluded, d_cfg, d_subs;
	rxgpu_devbuf d_mask, d_tmask, d_score, d_hist, d_slot_of, d_slots, d_sync;   // multi-term merge
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
// HIP event pair that cannot leak on an early error return
struct EventPair {
	hipEvent_t a = nullptr, b = nullptr;
	int create() {
		RX_HIP(hipEventCreate(&a));
		RX_HIP(hipEventCreate(&b));
		return RXGPU_OK;
	}
	float elapsed_ms() const {
		float ms = 0.f;
		(void)hipEventElapsedTime(&ms, a, b);
		return ms;
	}
	~EventPair() {
		if (a) (void)hipEventDestroy(a);
		if (b) (void)hipEventDestroy(b);
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
	for (rxgpu_devbuf* b : {&h->d_excluded, &h->d_cfg, &h->d_subs, &h->d_mask, &h->d_tmask, &h->d_score, &h->d_hist, &h->d_slot_of, &h->d_slots,
							&h->d_sync}) {
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

// Shared by both merges: slot state of the admitted documents (MergeInfo + MergerDocumentData), SoA, 8-byte members first
static int carve_slots(rxgpu_ft_index* h, size_t M, rxgpu::FtSlots& slots) {
	const size_t slots_bytes = M * (8 + 8 + 4 * 5 + 2 * 3 + 1) + 64;
	if (int rc = h->d_slots.ensure(slots_bytes); rc) return rc;
	char* sp = static_cast<char*>(h->d_slots.ptr);
	slots.last_ptr = reinterpret_cast<const uint64_t**>(sp);
	sp += M * 8;
	slots.next_ptr = reinterpret_cast<const uint64_t**>(sp);
	sp += M * 8;
	slots.doc = reinterpret_cast<uint32_t*>(sp);
	sp += M * 4;
	slots.proc = reinterpret_cast<float*>(sp);
	sp += M * 4;
	slots.rank = reinterpret_cast<float*>(sp);
	sp += M * 4;
	slots.last_cnt = reinterpret_cast<uint32_t*>(sp);
	sp += M * 4;
	slots.next_cnt = reinterpret_cast<uint32_t*>(sp);
	sp += M * 4;
	slots.switched_term = reinterpret_cast<uint16_t*>(sp);
	sp += M * 2;
	slots.last_counted = reinterpret_cast<uint16_t*>(sp);
	sp += M * 2;
	slots.terms_counter = reinterpret_cast<uint16_t*>(sp);
	sp += M * 2;
	slots.field = reinterpret_cast<uint8_t*>(sp);
	return RXGPU_OK;
}

static void fill_subterm(const rxgpu_ft_word& w, uint64_t total_docs, float proc, rxgpu::FtPosSubterm& ft) {
	ft.n = w.n;
	ft.doc = w.doc;
	ft.ent_off = w.ent_off;
	ft.ent_field = w.ent_field;
	ft.ent_tf = w.ent_tf;
	ft.ent_first_pos = w.ent_first_pos;
	ft.pos_off = w.pos_off;
	ft.fpos = w.fpos;
	// Bm25Rx::IDF(totalDocCount = totalNumDocs - 1, matchedDocCount = |postings|)  (bm25.h:19-26, mergerimpl.h:123-124, 203-205)
	const double td = double(total_docs - 1), md = double(w.n);
	double f = w.n ? std::log((td - md + 1) / md) / std::log(1 + td) : 0.2;
	if (f < 0.2) f = 0.2;
	ft.idf = f;
	ft.proc = proc;
	ft.gp_base = 0;
}

int rxgpu_ft_merge_simple_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_term_opts* opts, uint32_t nsub,
							  const uint32_t* word_ids, const float* procs, const uint8_t* excluded, uint32_t* out_doc, float* out_proc,
							  uint8_t* out_field, uint64_t cap, uint64_t* out_n) {
	RX_CHECK(h && cfg && opts && out_n, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: null argument");
	*out_n = 0;
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, "rxgpu_ft_merge_simple_raw: rxgpu_ft_set_docs was not called");
	if (nsub == 0) return RXGPU_OK;
	RX_CHECK(word_ids && procs && opts->field_boost && opts->need_sum_rank, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: null argument");
	{
		uint32_t nsum = 0;
		for (uint32_t f = 0; f < h->num_fields; ++f) nsum += opts->need_sum_rank[f] ? 1 : 0;
		RX_CHECK(nsum <= 8, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: more than 8 fields with needSumRank (GPU engine limit)");
	}
	std::lock_guard<std::mutex> lk(h->mtx);
	DevGuard dg(h->device);
	const uint32_t nf = h->num_fields;
	const uint64_t N = h->total_docs;
	std::vector<rxgpu::FtPosSubterm> subs(nsub);
	uint64_t total = 0, lookback_words = 0;
	uint32_t launches = 0;
	for (uint32_t s = 0; s < nsub; ++s) {
		auto it = h->words.find(word_ids[s]);
		RX_CHECK(it != h->words.end(), RXGPU_ERR_NOTFOUND, "rxgpu_ft_merge_simple_raw: unknown word id");
		fill_subterm(it->second, N, procs[s], subs[s]);
		total += subs[s].n;
		if (subs[s].n) {
			++launches;
			lookback_words += rxgpu::ft_pass_blocks(subs[s].n);
		}
	}
	RX_CHECK(total < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: more than 2^32 postings in one merge");
	const uint64_t max_merged = std::min<uint64_t>(cfg->merge_limit, total);   // Merge(): min(mergeLimit, totalORVids)
	if (max_merged == 0) return RXGPU_OK;
	RX_CHECK(cap >= max_merged && out_doc && out_proc && out_field, RXGPU_ERR_OVERFLOW, "rxgpu_ft_merge_simple_raw: output buffers too small");

	hipStream_t st = h->stream;
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
	if (int rc = h->d_cfg.ensure(fcfg.size() * sizeof(float) + nf); rc) return rc;
	RX_HIP(hipMemcpyAsync(h->d_cfg.ptr, fcfg.data(), fcfg.size() * sizeof(float), hipMemcpyHostToDevice, st));
	RX_HIP(hipMemcpyAsync(static_cast<char*>(h->d_cfg.ptr) + fcfg.size() * sizeof(float), opts->need_sum_rank, nf, hipMemcpyHostToDevice, st));
	const float* fc = static_cast<const float*>(h->d_cfg.ptr);

	// docsExcluded_ as the restricting mask (mergeSimple tests docsExcluded_[docId] || DocRemoved(docId), mergerimpl.h:209)
	const uint64_t nwords = (N + 31) / 32;
	if (int rc = h->d_mask.ensure(nwords * 4); rc) return rc;
	const uint8_t* d_excl = nullptr;
	if (excluded) {
		if (int rc = h->d_excluded.ensure(N); rc) return rc;
		RX_HIP(hipMemcpyAsync(h->d_excluded.ptr, excluded, N, hipMemcpyHostToDevice, st));
		d_excl = static_cast<const uint8_t*>(h->d_excluded.ptr);
	}
	rxgpu::launch_ft_mask_init(static_cast<uint32_t*>(h->d_mask.ptr), d_excl, N, st);

	const size_t sync_u32 = 8 + size_t(launches + 1) * 2;
	const size_t sync_bytes = ((sync_u32 * 4 + 7) & ~size_t(7)) + lookback_words * 8;
	if (int rc = h->d_sync.ensure(sync_bytes); rc) return rc;
	RX_HIP(hipMemsetAsync(h->d_sync.ptr, 0, sync_bytes, st));
	uint32_t* d_sync = static_cast<uint32_t*>(h->d_sync.ptr);
	uint32_t* d_error = d_sync;
	uint32_t* d_tickets = d_sync + 8;
	uint32_t* d_num_docs = d_tickets + launches + 1;
	unsigned long long* d_lookback = reinterpret_cast<unsigned long long*>(static_cast<char*>(h->d_sync.ptr) + ((sync_u32 * 4 + 7) & ~size_t(7)));
	if (int rc = h->d_slot_of.ensure(N * 4); rc) return rc;
	RX_HIP(hipMemsetAsync(h->d_slot_of.ptr, 0xFF, N * 4, st));
	rxgpu::FtSlots slots{};
	if (int rc = carve_slots(h, size_t(max_merged), slots); rc) return rc;

	EventPair ev;
	if (int rc = ev.create(); rc) return rc;
	RX_HIP(hipEventRecord(ev.a, st));
	uint32_t launch = 0;
	uint64_t lb_used = 0;
	for (uint32_t s = 0; s < nsub; ++s) {   // sub-terms in SortSubterms order; documents are unique inside one: a launch is race free
		if (!subs[s].n) continue;
		rxgpu::FtTermPass p{};
		p.cfg.num_fields = nf;
		p.cfg.words = h->d_words;
		p.cfg.avg_words = h->d_avg;
		p.cfg.k1 = cfg->bm25_k1;
		p.cfg.b = cfg->bm25_b;
		p.cfg.summation_ratio = cfg->summation_ranks_by_fields_ratio;
		p.cfg.opts_boost = opts->boost;
		p.cfg.term_len_boost_in = opts->term_len_boost;
		p.cfg.field_boost = fc + 0 * nf;
		p.cfg.bm25_boost = fc + 1 * nf;
		p.cfg.bm25_weight = fc + 2 * nf;
		p.cfg.term_len_boost = fc + 3 * nf;
		p.cfg.term_len_weight = fc + 4 * nf;
		p.cfg.position_boost = fc + 5 * nf;
		p.cfg.position_weight = fc + 6 * nf;
		p.cfg.need_sum_rank = reinterpret_cast<const uint8_t*>(fc + 7 * nf);
		p.sub = subs[s];
		p.slots = slots;
		p.mask = static_cast<const uint32_t*>(h->d_mask.ptr);
		p.removed = h->d_removed;
		p.slot_of = static_cast<uint32_t*>(h->d_slot_of.ptr);
		p.max_merged = uint32_t(max_merged);
		p.qp_idx = 1;
		p.simple = 1;
		p.num_docs_in = d_num_docs + launch;
		p.num_docs_out = d_num_docs + launch + 1;
		p.lookback = d_lookback + lb_used;
		p.ticket = d_tickets + launch;
		p.error_flag = d_error;
		rxgpu::launch_ft_term_pass(p, st);
		lb_used += rxgpu::ft_pass_blocks(subs[s].n);
		++launch;
	}
	RX_HIP(hipEventRecord(ev.b, st));
	RX_HIP(hipGetLastError());
	uint32_t tail[2] = {0, 0};
	RX_HIP(hipMemcpyAsync(&tail[0], d_num_docs + launch, 4, hipMemcpyDeviceToHost, st));
	RX_HIP(hipMemcpyAsync(&tail[1], d_error, 4, hipMemcpyDeviceToHost, st));
	RX_HIP(hipStreamSynchronize(st));
	const float ms = ev.elapsed_ms();
	h->stat_postings += total;
	h->stat_ms += ms;
	RX_CHECK(tail[1] == 0, RXGPU_ERR_DEVICE, "rxgpu_ft_merge_simple_raw: ordered look-back timed out on the device");
	const uint64_t n = tail[0];
	if (n) {
		RX_HIP(hipMemcpy(out_doc, slots.doc, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
		RX_HIP(hipMemcpy(out_proc, slots.proc, n * sizeof(float), hipMemcpyDeviceToHost));
		RX_HIP(hipMemcpy(out_field, slots.field, n, hipMemcpyDeviceToHost));
	}
	*out_n = n;
	return RXGPU_OK;
}

int rxgpu_ft_set_word_positions(rxgpu_ft_index* h, uint32_t word_id, uint64_t n, const uint32_t* doc, const uint32_t* pos_off, const uint64_t* fpos) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null ft index");
	RX_CHECK(n == 0 || (doc && pos_off && fpos), RXGPU_ERR_PARAMS, "rxgpu_ft_set_word_positions: null argument");
	// derive the (field, tf, first position) entries calcTermRankImpl groups out of IdRelType::Pos() (phrasemergerimpl.h:24-49)
	std::vector<uint32_t> ent_off(n + 1, 0), ent_tf, ent_first;
	std::vector<uint8_t> ent_field;
	for (uint64_t i = 0; i < n; ++i) {
		ent_off[i] = uint32_t(ent_field.size());
		RX_CHECK(pos_off[i + 1] > pos_off[i], RXGPU_ERR_PARAMS, "rxgpu_ft_set_word_positions: a posting without positions");
		for (uint32_t a = pos_off[i]; a < pos_off[i + 1];) {
			const uint32_t f = uint32_t(fpos[a] >> 56);
			RX_CHECK(f < h->num_fields, RXGPU_ERR_PARAMS, "rxgpu_ft_set_word_positions: field out of range");
			uint32_t b = a + 1;
			while (b < pos_off[i + 1] && uint32_t(fpos[b] >> 56) == f) ++b;
			ent_field.push_back(uint8_t(f));
			ent_tf.push_back(b - a);
			ent_first.push_back(uint32_t(fpos[a] & ((1u << 28) - 1)));
			a = b;
		}
	}
	ent_off[n] = uint32_t(ent_field.size());
	if (int rc = rxgpu_ft_set_word(h, word_id, n, doc, ent_off.data(), ent_field.data(), ent_tf.data(), ent_first.data()); rc) return rc;
	if (n == 0) return RXGPU_OK;
	std::lock_guard<std::mutex> lk(h->mtx);
	DevGuard dg(h->device);
	rxgpu_ft_word& w = h->words[word_id];
	if (int rc = upload(w.pos_off, pos_off, n + 1); rc) return rc;
	if (int rc = upload(w.fpos, fpos, size_t(pos_off[n])); rc) return rc;
	return RXGPU_OK;
}

int rxgpu_ft_merge_terms_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nterms, const int32_t* ops, const rxgpu_ft_term_opts* opts,
							 const uint32_t* sub_off, const uint32_t* word_ids, const float* procs, const uint8_t* excluded, uint32_t* out_doc,
							 float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter, uint64_t cap, uint64_t* out_n, int32_t* out_preselected) {
	RX_CHECK(h && cfg && out_n, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: null argument");
	*out_n = 0;
	if (out_preselected) *out_preselected = 0;
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, "rxgpu_ft_merge_terms_raw: rxgpu_ft_set_docs was not called");
	// QueryMergeData::Empty() (querymergedata.h:208)
	if (nterms == 0) return RXGPU_OK;
	RX_CHECK(ops && opts && sub_off, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: null argument");
	for (uint32_t t = 0; t < nterms; ++t) RX_CHECK(ops[t] >= 1 && ops[t] <= 3, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: op must be 1 (OR), 2 (AND) or 3 (NOT)");
	if (nterms == 1 && ops[0] == 3) return RXGPU_OK;
	RX_CHECK(nterms >= 2, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: a single-term query is Simple(): use rxgpu_ft_merge_simple_raw");
	RX_CHECK(nterms < 0xFFFF, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: too many terms");
	const uint32_t nsub_total = sub_off[nterms];
	RX_CHECK(nsub_total == 0 || (word_ids && procs), RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: null argument");

	std::lock_guard<std::mutex> lk(h->mtx);
	DevGuard dg(h->device);
	const uint32_t nf = h->num_fields;
	const uint64_t N = h->total_docs;
	std::vector<rxgpu::FtPosSubterm> subs(nsub_total);
	std::vector<uint64_t> term_postings(nterms, 0);
	uint64_t total_vids = 0;
	for (uint32_t t = 0; t < nterms; ++t) {
		uint64_t gp = 0;
		for (uint32_t s = sub_off[t]; s < sub_off[t + 1]; ++s) {
			auto it = h->words.find(word_ids[s]);
			RX_CHECK(it != h->words.end(), RXGPU_ERR_NOTFOUND, "rxgpu_ft_merge_terms_raw: unknown word id");
			const rxgpu_ft_word& w = it->second;
			RX_CHECK(w.n == 0 || w.fpos, RXGPU_ERR_LOGIC, "rxgpu_ft_merge_terms_raw: the word was uploaded without positions (rxgpu_ft_set_word_positions)");
			RX_CHECK(s == sub_off[t] || procs[s] <= procs[s - 1], RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: sub-terms must be sorted by proc, descending (SortSubterms)");
			rxgpu::FtPosSubterm& ft = subs[s];
			fill_subterm(w, N, procs[s], ft);
			ft.gp_base = gp;
			gp += w.n;
		}
		term_postings[t] = gp;
		total_vids += gp;   // totalORVids: MaxVDocs of every term (selecterimpl.h:546)
	}
	RX_CHECK(total_vids < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: more than 2^32 postings in one merge");
	const uint64_t max_merged = std::min<uint64_t>(cfg->merge_limit, total_vids);
	if (max_merged == 0) return RXGPU_OK;
	RX_CHECK(cap >= max_merged && out_doc && out_proc && out_field && out_terms_counter, RXGPU_ERR_OVERFLOW,
			 "rxgpu_ft_merge_terms_raw: output buffers too small");

	hipStream_t st = h->stream;
	// ---- configuration: 6 FTFieldConfig rows shared by all terms + per term (fieldBoost floats, needSumRank bytes)
	const size_t cfg_floats = size_t(6) * nf + size_t(nterms) * nf;
	std::vector<float> fcfg(cfg_floats);
	for (uint32_t f = 0; f < nf; ++f) {
		fcfg[0 * nf + f] = float(cfg->bm25_boost[f]);
		fcfg[1 * nf + f] = float(cfg->bm25_weight[f]);
		fcfg[2 * nf + f] = float(cfg->term_len_boost[f]);
		fcfg[3 * nf + f] = float(cfg->term_len_weight[f]);
		fcfg[4 * nf + f] = float(cfg->position_boost[f]);
		fcfg[5 * nf + f] = float(cfg->position_weight[f]);
	}
	std::vector<uint8_t> need_sum(size_t(nterms) * nf);
	for (uint32_t t = 0; t < nterms; ++t) {
		RX_CHECK(opts[t].field_boost && opts[t].need_sum_rank, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: null term options");
		uint32_t nsum = 0;
		for (uint32_t f = 0; f < nf; ++f) nsum += opts[t].need_sum_rank[f] ? 1 : 0;
		RX_CHECK(nsum <= 8, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: more than 8 fields with needSumRank (GPU engine limit)");
		for (uint32_t f = 0; f < nf; ++f) {
			fcfg[size_t(6 + t) * nf + f] = opts[t].field_boost[f];
			need_sum[size_t(t) * nf + f] = opts[t].need_sum_rank[f];
		}
	}
	if (int rc = h->d_cfg.ensure(cfg_floats * sizeof(float) + need_sum.size()); rc) return rc;
	RX_HIP(hipMemcpyAsync(h->d_cfg.ptr, fcfg.data(), cfg_floats * sizeof(float), hipMemcpyHostToDevice, st));
	RX_HIP(hipMemcpyAsync(static_cast<char*>(h->d_cfg.ptr) + cfg_floats * sizeof(float), need_sum.data(), need_sum.size(), hipMemcpyHostToDevice, st));
	const float* d_fc = static_cast<const float*>(h->d_cfg.ptr);
	const uint8_t* d_need_sum = reinterpret_cast<const uint8_t*>(d_fc + cfg_floats);
	if (int rc = h->d_subs.ensure(std::max<size_t>(1, subs.size()) * sizeof(rxgpu::FtPosSubterm)); rc) return rc;
	if (!subs.empty()) RX_HIP(hipMemcpyAsync(h->d_subs.ptr, subs.data(), subs.size() * sizeof(rxgpu::FtPosSubterm), hipMemcpyHostToDevice, st));
	const rxgpu::FtPosSubterm* d_subs = static_cast<const rxgpu::FtPosSubterm*>(h->d_subs.ptr);

	// ---- buildRestrictingBitmask
	const uint64_t nwords = (N + 31) / 32;
	if (int rc = h->d_mask.ensure(nwords * 4); rc) return rc;
	if (int rc = h->d_tmask.ensure(nwords * 4); rc) return rc;
	uint32_t* d_mask = static_cast<uint32_t*>(h->d_mask.ptr);
	uint32_t* d_tmask = static_cast<uint32_t*>(h->d_tmask.ptr);
	const uint8_t* d_excl = nullptr;
	if (excluded) {
		if (int rc = h->d_excluded.ensure(N); rc) return rc;
		RX_HIP(hipMemcpyAsync(h->d_excluded.ptr, excluded, N, hipMemcpyHostToDevice, st));
		d_excl = static_cast<const uint8_t*>(h->d_excluded.ptr);
	}
	rxgpu::launch_ft_mask_init(d_mask, d_excl, N, st);
	for (uint32_t t = 0; t < nterms; ++t) {
		if (ops[t] != 2) continue;
		RX_HIP(hipMemsetAsync(d_tmask, 0, nwords * 4, st));
		rxgpu::launch_ft_term_mask(d_subs + sub_off[t], sub_off[t + 1] - sub_off[t], term_postings[t], d_fc + size_t(6 + t) * nf, nf, d_tmask, st);
		rxgpu::launch_ft_mask_and(d_mask, d_tmask, nwords, st);
	}
	for (uint32_t t = 0; t < nterms; ++t) {
		if (ops[t] != 3) continue;
		rxgpu::launch_ft_mask_exclude(d_subs + sub_off[t], sub_off[t + 1] - sub_off[t], term_postings[t], d_mask, st);
	}

	// ---- synchronisation words: [0] error flag, [1] popcount, [2..3] preselect pick, then per launch: ticket + numDocs chain + look-back words
	uint32_t merge_launches = 0;
	uint64_t lookback_words = 0;
	for (uint32_t t = 0; t < nterms; ++t) {
		if (ops[t] == 3) continue;
		for (uint32_t s = sub_off[t]; s < sub_off[t + 1]; ++s) {
			if (!subs[s].n) continue;
			++merge_launches;
			lookback_words += rxgpu::ft_pass_blocks(subs[s].n);
		}
	}
	const uint64_t pre_blocks = (nwords + 255) / 256;
	const size_t sync_u32 = 8 + size_t(merge_launches + 1) * 2 + 2;   // header, (ticket, numDocs) per launch (+1), preselect ticket
	const size_t sync_bytes = ((sync_u32 * 4 + 7) & ~size_t(7)) + (lookback_words + pre_blocks) * 8;
	if (int rc = h->d_sync.ensure(sync_bytes); rc) return rc;
	RX_HIP(hipMemsetAsync(h->d_sync.ptr, 0, sync_bytes, st));
	uint32_t* d_sync = static_cast<uint32_t*>(h->d_sync.ptr);
	uint32_t* d_error = d_sync + 0;
	uint32_t* d_pop = d_sync + 1;
	uint32_t* d_pick = d_sync + 2;
	uint32_t* d_pre_ticket = d_sync + 4;
	uint32_t* d_tickets = d_sync + 8;                        // [merge_launches]
	uint32_t* d_num_docs = d_tickets + merge_launches + 1;   // [merge_launches + 1]
	unsigned long long* d_lookback = reinterpret_cast<unsigned long long*>(static_cast<char*>(h->d_sync.ptr) + ((sync_u32 * 4 + 7) & ~size_t(7)));

	// ---- estimateNumDocsInMerge (merger.h:239-267) and the 2-phase gate (mergerimpl.h:486-490)
	bool preselected = false;
	{
		uint64_t est_or = 0, est_and = UINT64_MAX;
		for (uint32_t t = 0; t < nterms; ++t) {
			if (ops[t] == 3) continue;
			if (ops[t] == 2) {
				est_and = std::min(est_and, term_postings[t]);
			} else {
				est_or += term_postings[t];
			}
		}
		const uint64_t est = std::min(std::min(est_or, est_and), N);
		if (est > cfg->merge_limit && N > cfg->merge_limit) {
			rxgpu::launch_ft_mask_popcount(d_mask, nwords, d_pop, st);
			uint32_t pop = 0;
			RX_HIP(hipMemcpyAsync(&pop, d_pop, 4, hipMemcpyDeviceToHost, st));
			RX_HIP(hipStreamSynchronize(st));
			preselected = pop > cfg->merge_limit;
		}
	}
	if (preselected) {   // preselectMostRelevantDocs (mergerimpl.h:386-464)
		if (int rc = h->d_score.ensure(N * 2); rc) return rc;
		if (int rc = h->d_hist.ensure(65536 * 4); rc) return rc;
		RX_HIP(hipMemsetAsync(h->d_score.ptr, 0, N * 2, st));
		RX_HIP(hipMemsetAsync(h->d_hist.ptr, 0, 65536 * 4, st));
		uint16_t* d_score = static_cast<uint16_t*>(h->d_score.ptr);
		for (uint32_t t = 0; t < nterms; ++t) {
			if (ops[t] == 3) continue;
			RX_HIP(hipMemsetAsync(d_tmask, 0, nwords * 4, st));
			bool same = true;
			for (uint32_t f = 0; f < nf; ++f) same = same && opts[t].field_boost[f] == opts[t].field_boost[0];
			for (uint32_t s = sub_off[t]; s < sub_off[t + 1]; ++s) {
				rxgpu::launch_ft_prescore(subs[s], d_mask, d_tmask, d_score, d_fc + size_t(6 + t) * nf, nf, same, opts[t].boost, st);
			}
		}
		rxgpu::FtPreselect ps{};
		ps.mask_in = d_mask;
		ps.mask = d_mask;
		ps.term_mask = d_tmask;
		ps.score = d_score;
		ps.hist = static_cast<uint32_t*>(h->d_hist.ptr);
		ps.pick = d_pick;
		ps.total_docs = N;
		ps.removed = h->d_removed;
		ps.max_merged = uint32_t(max_merged);
		ps.lookback = d_lookback + lookback_words;
		ps.ticket = d_pre_ticket;
		ps.error_flag = d_error;
		rxgpu::launch_ft_preselect(ps, st);
	}

	// ---- mergeTerm for every term that is not a NOT
	if (int rc = h->d_slot_of.ensure(N * 4); rc) return rc;
	RX_HIP(hipMemsetAsync(h->d_slot_of.ptr, 0xFF, N * 4, st));
	rxgpu::FtSlots slots{};
	if (int rc = carve_slots(h, size_t(max_merged), slots); rc) return rc;

	EventPair ev;
	if (int rc = ev.create(); rc) return rc;
	RX_HIP(hipEventRecord(ev.a, st));
	uint32_t launch = 0;
	uint64_t lb_used = 0, merged_postings = 0;
	uint16_t qp = 0;
	for (uint32_t t = 0; t < nterms; ++t) {
		if (ops[t] == 3) continue;
		++qp;
		for (uint32_t s = sub_off[t]; s < sub_off[t + 1]; ++s) {
			if (!subs[s].n) continue;
			rxgpu::FtTermPass p{};
			p.cfg.num_fields = nf;
			p.cfg.words = h->d_words;
			p.cfg.avg_words = h->d_avg;
			p.cfg.k1 = cfg->bm25_k1;
			p.cfg.b = cfg->bm25_b;
			p.cfg.summation_ratio = cfg->summation_ranks_by_fields_ratio;
			p.cfg.opts_boost = opts[t].boost;
			p.cfg.term_len_boost_in = opts[t].term_len_boost;
			p.cfg.field_boost = d_fc + size_t(6 + t) * nf;
			p.cfg.need_sum_rank = d_need_sum + size_t(t) * nf;
			p.cfg.bm25_boost = d_fc + 0 * nf;
			p.cfg.bm25_weight = d_fc + 1 * nf;
			p.cfg.term_len_boost = d_fc + 2 * nf;
			p.cfg.term_len_weight = d_fc + 3 * nf;
			p.cfg.position_boost = d_fc + 4 * nf;
			p.cfg.position_weight = d_fc + 5 * nf;
			p.sub = subs[s];
			p.slots = slots;
			p.mask = d_mask;
			p.removed = preselected ? nullptr : h->d_removed;   // needToCheckRemoved_ = false after the preselect
			p.slot_of = static_cast<uint32_t*>(h->d_slot_of.ptr);
			p.max_merged = uint32_t(max_merged);
			p.qp_idx = qp;
			p.distance_weight = float(cfg->distance_weight);
			p.distance_boost = float(cfg->distance_boost);
			p.num_docs_in = d_num_docs + launch;
			p.num_docs_out = d_num_docs + launch + 1;
			p.lookback = d_lookback + lb_used;
			p.ticket = d_tickets + launch;
			p.error_flag = d_error;
			rxgpu::launch_ft_term_pass(p, st);
			lb_used += rxgpu::ft_pass_blocks(subs[s].n);
			merged_postings += subs[s].n;
			++launch;
		}
	}
	RX_HIP(hipEventRecord(ev.b, st));
	RX_HIP(hipGetLastError());
	uint32_t tail[2] = {0, 0};   // numDocs, error flag
	RX_HIP(hipMemcpyAsync(&tail[0], d_num_docs + launch, 4, hipMemcpyDeviceToHost, st));
	RX_HIP(hipMemcpyAsync(&tail[1], d_error, 4, hipMemcpyDeviceToHost, st));
	RX_HIP(hipStreamSynchronize(st));
	const float ms = ev.elapsed_ms();
	h->stat_postings += merged_postings;
	h->stat_ms += ms;
	RX_CHECK(tail[1] == 0, RXGPU_ERR_DEVICE, "rxgpu_ft_merge_terms_raw: ordered look-back timed out on the device");
	const uint64_t n = tail[0];
	if (n) {
		RX_HIP(hipMemcpy(out_doc, slots.doc, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
		RX_HIP(hipMemcpy(out_proc, slots.proc, n * sizeof(float), hipMemcpyDeviceToHost));
		RX_HIP(hipMemcpy(out_field, slots.field, n, hipMemcpyDeviceToHost));
		RX_HIP(hipMemcpy(out_terms_counter, slots.terms_counter, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
	}
	*out_n = n;
	if (out_preselected) *out_preselected = preselected ? 1 : 0;
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
