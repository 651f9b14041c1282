// C-ABI of the BM25 merge (include/rxgpu.h, rxgpu_ft_*): device mirror of the ft_fast posting lists + the scoring launch.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <memory>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/rxgpu.h"
#include "rccl_dyn.h"
#include "rxgpu_internal.h"
#include "ft_phrase_cut.h"
#include "ft_rank.hip.h"

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
	uint32_t* range_off = nullptr; // [n_ranges]: first posting with doc >= k * kFtRangeDocs (ft_ranges finds its segment of the list here)
	uint32_t n_ranges = 0;
	uint32_t last_doc = 0;         // largest document id of the list: checked against total_docs when a merge uses the word
	uint64_t df = 0;               // document-range shards: the word's document frequency over the WHOLE index (this list is a fragment); 0: n
	uint64_t* fpos = nullptr;
	std::shared_ptr<void> pool;    // set for words decoded on the device (rxgpu_ft_set_words_packed): the arrays are slices of one allocation
	void release() {
		if (!pool) {
			for (void* p : {static_cast<void*>(doc), static_cast<void*>(ent_off), static_cast<void*>(ent_field), static_cast<void*>(ent_tf),
							static_cast<void*>(ent_first_pos), static_cast<void*>(pos_off), static_cast<void*>(fpos), static_cast<void*>(range_off)}) {
				if (p) (void)hipFree(p);
			}
		}
		*this = rxgpu_ft_word{};
	}
};

struct rxgpu_ft_shard_set;
struct rxgpu_ft_index {
	int device = 0;
	uint32_t num_fields = 0;
	uint64_t total_docs = 0;
	// Document-range shards (rxgpu_ft_create_sharded, SURVEY 8e "BM25").  The handle the caller holds owns the shards (shard_set); a shard is
	// an ordinary index over the GLOBAL document space that merges its own ranges only (sh_*: set by the sharded layer around every merge).
	rxgpu_ft_shard_set* shard_set = nullptr;
	uint32_t sh_range_begin = 0, sh_range_count = 0, sh_index = 0, sh_total = 0;
	const uint32_t* sh_hist = nullptr;   // every shard's folded histogram as gathered on this shard's device
	const uint32_t* sh_pos = nullptr;    // shard -> position in the gathered buffers
	float* d_words = nullptr;
	float* d_avg = nullptr;
	uint8_t* d_removed = nullptr;
	uint32_t* d_removed_bits = nullptr;   // the same as one bit per document (the sparse train, ft_sparse.hip); null: no document is removed
	std::vector<float> h_avg;             // avg_words as uploaded (the sparse train's eligibility test reads it)
	std::atomic<uint64_t> trains_dense{0}, trains_sparse{0};   // merges by launch train (rxgpu_ft_read_train_stats)
	std::unordered_map<uint32_t, rxgpu_ft_word> words;
	std::mutex mtx;
	// Concurrent merges (several planner threads query one index at a time): extra LANES — own stream, scratch, staging, events — behind
	// the same dictionary.  A lane is a rxgpu_ft_index whose `root` points at the handle that owns words and statistics; the handle
	// itself is lane 0 and the only one the resident / hybrid calls use.  Merges hold dict_mtx shared, dictionary updates exclusively.
	rxgpu_ft_index* root = nullptr;
	std::vector<std::unique_ptr<rxgpu_ft_index>> lanes;
	std::mutex lanes_mtx;
	std::shared_mutex dict_mtx;
	std::atomic<uint32_t> next_lane{0};
	// Q merges in ONE launch train (rxgpu_ft_merge_batch_raw): a scratch set per query of the batch (lanes without a stream of their own: the
	// whole train runs on batch_stream), the Q FtPlan structs back to back in HBM + their pinned staging, events around the train
	std::vector<std::unique_ptr<rxgpu_ft_index>> batch_lanes;
	std::mutex batch_mtx;
	hipStream_t batch_stream = nullptr;
	rxgpu_devbuf d_batch_plans;
	void* h_batch_plans = nullptr;
	hipEvent_t ev_ba = nullptr, ev_bb = nullptr;
	uint64_t batch_trains = 0, batch_merges = 0;
	const std::unordered_map<uint32_t, rxgpu_ft_word>& dict() const { return root ? root->words : words; }
	hipStream_t stream = nullptr;
	rxgpu_devbuf d_state, d_out;   // per-merge scratch (plan + tables) and the packed result
	rxgpu_devbuf d_excl;           // docsExcluded of the running merge
	rxgpu_devbuf d_areas;          // MergeDataAreas: per merged document and field {held, insertions} + the areas themselves
	rxgpu_devbuf d_pk_in, d_pk_cnt, d_pk_segs, d_pk_outs;   // rxgpu_ft_set_words_packed: streams + offsets, counts, pieces, slices (kept and grown)
	hipStream_t pk_streams[4] = {nullptr, nullptr, nullptr, nullptr};   // ... and the streams its chunked counting pass runs on (created on first use)
	std::vector<rxgpu_devbuf> d_phrase_a, d_phrase_b;   // per phrase of a query: plan + admission slots, workspace + the packed rows
	hipEvent_t ev_pha = nullptr, ev_phb = nullptr;      // around the phrase kernels
	// tables every merge finds ZEROED and leaves zeroed (the kernel that reads one last clears it): pre-score histogram, look-back words of
	// the preselect, bucket counters, synchronisation words, the occupancy (rank) plane of the entry rows.  Cleared by the host only when (re)allocated or after a failed merge.
	rxgpu_devbuf d_clean;
	uint64_t clean_docs = 0;
	bool clean_dirty = true;
	void* h_pinned = nullptr;     // staging: plan upload / result download
	size_t h_pinned_bytes = 0;
	hipEvent_t ev_a = nullptr, ev_b = nullptr;
	int ensure_pinned(size_t need) {
		if (need <= h_pinned_bytes) return RXGPU_OK;
		if (h_pinned) (void)hipHostFree(h_pinned);
		h_pinned = nullptr;
		h_pinned_bytes = 0;
		const size_t want = need + need / 2 + 4096;
		if (hipHostMalloc(&h_pinned, want, hipHostMallocDefault) != hipSuccess) {
			set_error("hipHostMalloc failed");
			return RXGPU_ERR_NOMEM;
		}
		h_pinned_bytes = want;
		return RXGPU_OK;
	}
	// a merge left in HBM for the hybrid fusion (rxgpu_ft_merge_*_resident): no export, no wait; checked by finish_pending()
	// The steps of one hybrid query (resident merge, prepare, fuse) each take `mtx` on their own, so the result is guarded by a SESSION: opened
	// by the resident merge for the calling thread, closed by that thread's fusion.  While it is open ordinary merges keep off this lane
	// (checkout_lane), other threads' resident merges wait on res_cv; a session nobody fuses is taken over after kResidentPatience and its
	// owner's later calls fail with RXGPU_ERR_LOGIC (generation mismatch) instead of reading another query's result.
	bool res_session = false;
	std::thread::id res_owner;
	uint64_t res_generation = 0;
	std::condition_variable res_cv;
	bool res_pending = false;
	bool res_has_syn = false;   // the resident merge had multi-word synonyms: its terms counters carry the 0xFFFF marks of the removed documents
	uint32_t res_cap = 0;          // max_merged of that merge (the packed layout of d_out depends on it)
	bool prep_done = false;        // hybrid_prepare_kernel has been enqueued behind that merge (with prep_sig's reranker / min_rank)
	double prep_sig[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	rxgpu_devbuf d_fuse;           // fusion scratch: radix ping-pong keys / classes
	hipEvent_t ev_knn = nullptr;   // orders the fusion behind the KNN search's stream
	hipEvent_t ev_fa = nullptr, ev_fb = nullptr;   // around the join kernel (rxgpu_hybrid_read_stats)
	hipEvent_t ev_pa = nullptr, ev_pb = nullptr;   // around the prepare kernel
	bool prep_timed = false;
	double prep_ms = 0.0;
	uint64_t fuse_calls = 0;
	double fuse_ms = 0.0;
	double fuse_stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // RXGPU_FUSE_STAMPS: summed phase stamps of the fusion kernel (us since its first)
	double packed_wall_ms = 0.0;                            // rxgpu_ft_set_words_packed*: wall time inside the calls (rxgpu_ft_read_packed_wall)
	double packed_count_ms = 0.0, packed_write_ms = 0.0;   // rxgpu_ft_set_words_packed: device time of the two decode kernels ...
	uint64_t packed_bytes_in = 0, packed_bytes_out = 0;    // ... the stream bytes they read and the array bytes they wrote
	uint64_t stat_postings = 0;
	double stat_ms = 0.0;
	double stamps[64] = {};   // RXGPU_FT_STAMPS: summed phase stamps (relative to the workgroup's first), see rxgpu_ft_read_stats
	double trace_us[6] = {0, 0, 0, 0, 0, 0};   // RXGPU_FT_TRACE: plan build, staging + upload, launches, wait + download, unpack, merges
};

// ---------------------------------------------------------------------------------------------- document-range shards (SURVEY 8e "BM25")
// "Shard by doc-id range (each GPU holds the posting fragments of its docs; idf uses global N and df ...); exchange = ... the uint16 pre-score
// histogram for the global threshold".  The index is cut into contiguous runs of 8192-document ranges, one run per listed device (a device
// may repeat).  Every shard is an ordinary rxgpu_ft_index over the GLOBAL document space — the per-document statistics are replicated (a
// few bytes per document), the posting lists, the bulk, are split: a shard holds the fragment of every list that falls into its documents,
// with the whole list's length as document frequency — and runs the ordinary kernels over its own ranges.  The merge algorithm is
// range-parallel with three per-query facts that span the ranges; between the kernels exactly those travel, over RCCL when the library is
// there (one all-gather each; rccl_dyn.h), on the streams, without a host round trip:
//   behind ft_ranges   every shard's folded pre-score histogram + the popcount of its mask words  (266 KB per shard)
//                      -> the 2-phase gate and preselectMostRelevantDocs' threshold (mergerimpl.h:386-464, 486-490) are decided on the sums,
//                         the ties kept at the threshold score are handed out in document order = shard order
//   behind ft_adders   every shard's table of documents first met per (sub-term row, range)      (rows x ranges x 4 B per shard)
//                      -> the sum is the table of the whole index: the merge slot of every document (addDoc order, merger.h:161-180) and the
//                         cut at maxMergedDocs are the single index's
// so every shard writes its documents at their GLOBAL merge slots, and the caller's list is the slot-wise union: the single handle's result,
// bit for bit (tests/test_gpu_ft_sharded.py).  postProcessResults' maximum (merger.h:111-155) is taken by the host merger over that list.
struct rxgpu_ft_shard_set {
	std::vector<rxgpu_ft_index*> shards;
	std::vector<int> devices;
	uint32_t n_ranges = 0;                  // of the whole index; 0: rxgpu_ft_set_docs has not run
	uint32_t per = 0;                       // ranges per shard of the current cut (the last shard also takes what lies behind S * per)
	// the exchange: one RCCL rank per DISTINCT device, a device's shards are `slots` consecutive pieces of its rank's buffers
	uint32_t nranks = 0, slots = 0;
	std::vector<int> rank_dev;
	std::vector<uint32_t> shard_rank, shard_slot, pos;   // pos[s] = rank * slots + slot: where shard s lies in a gathered buffer
	std::shared_ptr<rxgpu::RcclCommSet> cs; // the process-wide communicators over rank_dev when the shards span several devices (rccl_dyn.h); else null
	bool host_exchange = false;             // RXGPU_SHARD_MERGE=host, or several devices without RCCL (note says why): the pieces travel through the host
	std::string note;
	std::vector<hipStream_t> rstream;       // per rank
	std::vector<hipEvent_t> ev_shard, ev_rank;
	std::vector<uint32_t*> d_pos;           // per rank: pos[] on the device
	std::vector<rxgpu_devbuf> d_send[2], d_recv[2];   // per rank; [0] histograms, [1] adder tables
	uint64_t collectives = 0, merges = 0;
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
inline uint64_t word_df(const rxgpu_ft_word& w) { return w.df ? w.df : w.n; }
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

namespace {
void ft_shards_destroy(rxgpu_ft_shard_set* ss);
}

int rxgpu_ft_create_sharded(uint32_t num_fields, uint32_t n_devices, const int* devices, rxgpu_ft_index** out) {
	RX_CHECK(out && devices && n_devices >= 1 && n_devices <= 64, RXGPU_ERR_PARAMS, "rxgpu_ft_create_sharded: bad arguments (1..64 devices)");
	*out = nullptr;
	int prev = -1;
	(void)hipGetDevice(&prev);
	rxgpu_ft_index* h = nullptr;
	if (int rc = rxgpu_ft_create(num_fields, devices[0], &h); rc) return rc;   // the handle the caller holds: no dictionary of its own
	auto* ss = new rxgpu_ft_shard_set();
	h->shard_set = ss;
	auto fail = [&](int rc) {
		const std::string msg = rxgpu_last_error();
		rxgpu_ft_destroy(h);
		if (prev >= 0) (void)hipSetDevice(prev);
		set_error(msg);
		return rc;
	};
	std::vector<uint32_t> per_rank;
	for (uint32_t s = 0; s < n_devices; ++s) {
		rxgpu_ft_index* sh = nullptr;
		if (int rc = rxgpu_ft_create(num_fields, devices[s], &sh); rc) return fail(rc);
		ss->shards.push_back(sh);
		ss->devices.push_back(devices[s]);
		uint32_t r = 0;
		while (r < ss->rank_dev.size() && ss->rank_dev[r] != devices[s]) ++r;
		if (r == ss->rank_dev.size()) {
			ss->rank_dev.push_back(devices[s]);
			per_rank.push_back(0);
		}
		ss->shard_rank.push_back(r);
		ss->shard_slot.push_back(per_rank[r]++);
	}
	ss->nranks = uint32_t(ss->rank_dev.size());
	ss->slots = *std::max_element(per_rank.begin(), per_rank.end());
	for (uint32_t s = 0; s < n_devices; ++s) ss->pos.push_back(ss->shard_rank[s] * ss->slots + ss->shard_slot[s]);
	ss->rstream.assign(ss->nranks, nullptr);
	ss->ev_rank.assign(ss->nranks, nullptr);
	ss->d_pos.assign(ss->nranks, nullptr);
	ss->ev_shard.assign(n_devices, nullptr);
	for (int k = 0; k < 2; ++k) {
		ss->d_send[k].resize(ss->nranks);
		ss->d_recv[k].resize(ss->nranks);
	}
	for (uint32_t r = 0; r < ss->nranks; ++r) {
		hipError_t e = hipSetDevice(ss->rank_dev[r]);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&ss->rstream[r], hipStreamNonBlocking);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&ss->ev_rank[r], hipEventDisableTiming);
		if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ss->d_pos[r]), ss->pos.size() * sizeof(uint32_t));
		if (e == hipSuccess) e = hipMemcpy(ss->d_pos[r], ss->pos.data(), ss->pos.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
		if (e != hipSuccess) {
			set_error(std::string("rxgpu_ft_create_sharded: device ") + std::to_string(ss->rank_dev[r]) + ": " + hipGetErrorString(e));
			return fail(RXGPU_ERR_DEVICE);
		}
	}
	for (uint32_t s = 0; s < n_devices; ++s) {
		hipError_t e = hipSetDevice(devices[s]);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&ss->ev_shard[s], hipEventDisableTiming);
		if (e != hipSuccess) {
			set_error(std::string("rxgpu_ft_create_sharded: ") + hipGetErrorString(e));
			return fail(RXGPU_ERR_DEVICE);
		}
	}
	// the exchange: RXGPU_SHARD_MERGE=host -> through the host; one device -> copies on that device; several devices -> one RCCL
	// communicator over them (opened on demand; missing / failing: through the host, one line on stderr)
	const char* mode = getenv("RXGPU_SHARD_MERGE");
	if (mode && std::strcmp(mode, "host") == 0) {
		ss->host_exchange = true;
		ss->note = "RXGPU_SHARD_MERGE=host";
	} else if (ss->nranks > 1) {
		ss->cs = rxgpu::rccl_comm_set(ss->rank_dev, &ss->note);
		if (!ss->cs) {
			ss->host_exchange = true;
			fprintf(stderr, "rxgpu: sharded ft index over %u device slot(s): %s — the shards' histograms and tables travel through the host\n", n_devices, ss->note.c_str());
		}
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	*out = h;
	return RXGPU_OK;
}
uint32_t rxgpu_ft_shard_count(const rxgpu_ft_index* h) { return h && h->shard_set ? uint32_t(h->shard_set->shards.size()) : 0; }
// ranges of the fullest shard / ranges of an even cut (1.0: even; an index that grew by step commits piles its new ranges on the last shard)
double rxgpu_ft_shard_imbalance(const rxgpu_ft_index* h) {
	if (!h || !h->shard_set || !h->shard_set->n_ranges) return 1.0;
	const rxgpu_ft_shard_set* ss = h->shard_set;
	uint32_t most = 0;
	for (const rxgpu_ft_index* sh : ss->shards) most = std::max(most, sh->sh_range_count);
	const double even = double(ss->n_ranges) / double(ss->shards.size());
	return even > 0 ? std::max(1.0, double(most) / std::max(1.0, even)) : 1.0;
}
int rxgpu_ft_shard_exchange_mode(const rxgpu_ft_index* h) { return h && h->shard_set ? (h->shard_set->host_exchange ? 0 : 1) : -1; }
uint64_t rxgpu_ft_shard_collectives(const rxgpu_ft_index* h) { return h && h->shard_set ? h->shard_set->collectives : 0; }
int rxgpu_ft_shard_ranges(const rxgpu_ft_index* h, uint32_t shard, uint32_t* range_begin, uint32_t* range_count) {
	RX_CHECK(h && h->shard_set && shard < h->shard_set->shards.size() && range_begin && range_count, RXGPU_ERR_PARAMS, "rxgpu_ft_shard_ranges: bad arguments");
	*range_begin = h->shard_set->shards[shard]->sh_range_begin;
	*range_count = h->shard_set->shards[shard]->sh_range_count;
	return RXGPU_OK;
}

namespace {
void release_lane(rxgpu_ft_index* h) {   // what a lane owns: stream, scratch, staging, events
	for (rxgpu_devbuf* b : {&h->d_state, &h->d_out, &h->d_clean, &h->d_fuse, &h->d_excl, &h->d_areas, &h->d_pk_in, &h->d_pk_cnt, &h->d_pk_segs, &h->d_pk_outs}) b->release();
	for (rxgpu_devbuf& b : h->d_phrase_a) b.release();
	for (rxgpu_devbuf& b : h->d_phrase_b) b.release();
	for (hipEvent_t e : {h->ev_knn, h->ev_fa, h->ev_fb, h->ev_pa, h->ev_pb, h->ev_pha, h->ev_phb}) {
		if (e) (void)hipEventDestroy(e);
	}
	for (hipStream_t& ps : h->pk_streams) {
		if (ps) (void)hipStreamDestroy(ps);
		ps = nullptr;
	}
	if (h->h_pinned) (void)hipHostFree(h->h_pinned);
	if (h->ev_a) (void)hipEventDestroy(h->ev_a);
	if (h->ev_b) (void)hipEventDestroy(h->ev_b);
	if (h->stream) (void)hipStreamDestroy(h->stream);
}

constexpr uint32_t kFtMaxLanes = 16;
uint32_t ft_lane_limit() {
	static const uint32_t v = [] {
		const char* e = std::getenv("RXGPU_FT_LANES");
		const long n = e && *e ? std::atol(e) : 4;
		return uint32_t(n < 1 ? 1 : (n > long(kFtMaxLanes) ? long(kFtMaxLanes) : n));
	}();
	return v;
}

// A free lane for one merge, locked; then the dictionary, shared.  The first free one of: the handle itself, the lanes made so far, a new
// lane (up to RXGPU_FT_LANES, default 4); all busy: wait for one in turn.
struct LaneLock {
	rxgpu_ft_index* lane = nullptr;
	std::unique_lock<std::mutex> lk;
	std::shared_lock<std::shared_mutex> dict;
};
int checkout_lane(rxgpu_ft_index* h, LaneLock& out) {
	if (h->shard_set) {   // a sharded index runs one merge at a time: every shard's handle is busy with it
		out.lane = h;
		out.lk = std::unique_lock<std::mutex>(h->mtx);
		out.dict = std::shared_lock<std::shared_mutex>(h->dict_mtx);
		return RXGPU_OK;
	}
	auto take = [&](rxgpu_ft_index* l, std::unique_lock<std::mutex>&& lk) {
		out.lane = l;
		out.lk = std::move(lk);
		out.dict = std::shared_lock<std::shared_mutex>(h->dict_mtx);
		if (l != h) {   // what a merge reads of the statistics
			l->total_docs = h->total_docs;
			l->d_words = h->d_words;
			l->d_avg = h->d_avg;
			l->d_removed = h->d_removed;
			l->d_removed_bits = h->d_removed_bits;
			l->h_avg = h->h_avg;
		}
	};
	{
		std::unique_lock<std::mutex> lk(h->mtx, std::try_to_lock);
		if (lk.owns_lock() && !h->res_session) {   // (a resident merge parked on the handle: ordinary merges take the other lanes)
			take(h, std::move(lk));
			return RXGPU_OK;
		}
	}
	std::vector<rxgpu_ft_index*> have;
	{
		std::lock_guard<std::mutex> g(h->lanes_mtx);
		for (auto& l : h->lanes) have.push_back(l.get());
	}
	for (rxgpu_ft_index* l : have) {
		std::unique_lock<std::mutex> lk(l->mtx, std::try_to_lock);
		if (lk.owns_lock()) {
			take(l, std::move(lk));
			return RXGPU_OK;
		}
	}
	if (have.size() + 1 < std::max<uint32_t>(ft_lane_limit(), 2)) {   // (at least one lane besides the handle: a parked resident merge keeps the handle busy)
		auto lane = std::make_unique<rxgpu_ft_index>();
		lane->device = h->device;
		lane->num_fields = h->num_fields;
		lane->root = h;
		DevGuard dg(h->device);
		if (hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking) != hipSuccess) {
			set_error("hipStreamCreateWithFlags failed");
			return RXGPU_ERR_DEVICE;
		}
		rxgpu_ft_index* l = lane.get();
		std::unique_lock<std::mutex> lk(l->mtx);
		{
			std::lock_guard<std::mutex> g(h->lanes_mtx);
			h->lanes.push_back(std::move(lane));
		}
		take(l, std::move(lk));
		return RXGPU_OK;
	}
	uint32_t turn = h->next_lane.fetch_add(1) % uint32_t(have.size() + 1);
	if (turn == 0) {
		std::unique_lock<std::mutex> lk(h->mtx);
		if (!h->res_session) {
			take(h, std::move(lk));
			return RXGPU_OK;
		}
		turn = 1;   // (have is not empty: the branch above makes a second lane before anyone queues)
	}
	rxgpu_ft_index* l = have[turn - 1];
	take(l, std::unique_lock<std::mutex>(l->mtx));
	return RXGPU_OK;
}
}  // namespace

// ---- the sharded handle's side of the dictionary calls (the merge itself: run_merge_sharded)
static int ft_shards_set_docs(rxgpu_ft_index* h, uint64_t total_docs, const float* words_in_field, const float* avg_words, const uint8_t* removed) {
	rxgpu_ft_shard_set* ss = h->shard_set;
	std::lock_guard<std::mutex> lk(h->mtx);
	const uint32_t S = uint32_t(ss->shards.size());
	const uint32_t n_ranges = uint32_t((total_docs + rxgpu::kFtRangeDocs - 1) / rxgpu::kFtRangeDocs);
	// The cut: shard s starts at range s * per.  It is fixed by the first rxgpu_ft_set_docs and KEPT while any shard holds words — the index
	// grows through step commits (IndexText::commitFulltextImpl calls this with a larger totalDocs and re-uploads only the changed words),
	// and the fragments already on the shards must stay where the cut put them: new ranges go to the last shard, an even cut comes back
	// with the next index built from scratch (rxgpu_ft_shard_imbalance tells the caller when that is worth it).
	bool holds_words = false;
	for (rxgpu_ft_index* sh : ss->shards) holds_words = holds_words || !sh->words.empty();
	if (!ss->per || !holds_words) ss->per = std::max<uint32_t>(1, (n_ranges + S - 1) / S);
	const uint32_t per = ss->per;
	for (uint32_t s = 0; s < S; ++s) {
		rxgpu_ft_index* sh = ss->shards[s];
		if (int rc = rxgpu_ft_set_docs(sh, total_docs, words_in_field, avg_words, removed); rc) return rc;   // replicated: a few bytes per document
		sh->sh_index = s;
		sh->sh_total = S;
		sh->sh_range_begin = std::min(s * per, n_ranges);
		sh->sh_range_count = s + 1 == S ? n_ranges - sh->sh_range_begin : std::min(per, n_ranges - sh->sh_range_begin);
	}
	ss->n_ranges = n_ranges;
	h->total_docs = total_docs;
	return RXGPU_OK;
}

// One dictionary word: every shard takes the postings of ITS documents (ids stay global) and the whole list's length as document frequency.
// Either the flat form (ent_*) or the positions form (pos_off / fpos).
static int ft_shards_set_word(rxgpu_ft_index* h, uint32_t word_id, uint64_t n, const uint32_t* doc, const uint32_t* ent_off, const uint8_t* ent_field,
							  const uint32_t* ent_tf, const uint32_t* ent_first_pos, const uint32_t* pos_off, const uint64_t* fpos) {
	rxgpu_ft_shard_set* ss = h->shard_set;
	std::lock_guard<std::mutex> lk(h->mtx);
	RX_CHECK(ss->n_ranges > 0, RXGPU_ERR_LOGIC, "a sharded ft index cuts its posting lists at the document ranges: call rxgpu_ft_set_docs first");
	for (uint64_t i = 1; i < n; ++i) RX_CHECK(doc[i] > doc[i - 1], RXGPU_ERR_PARAMS, "rxgpu_ft_set_word: document ids must ascend strictly");
	RX_CHECK(n == 0 || doc[n - 1] < h->total_docs, RXGPU_ERR_PARAMS, "rxgpu_ft_set_word: a posting list holds a document id >= total_docs (rxgpu_ft_set_docs)");
	for (rxgpu_ft_index* sh : ss->shards) {
		const uint64_t d_lo = uint64_t(sh->sh_range_begin) * rxgpu::kFtRangeDocs, d_hi = d_lo + uint64_t(sh->sh_range_count) * rxgpu::kFtRangeDocs;
		const uint64_t a = uint64_t(std::lower_bound(doc, doc + n, d_lo, [](uint32_t x, uint64_t v) { return uint64_t(x) < v; }) - doc);
		const uint64_t b = uint64_t(std::lower_bound(doc, doc + n, d_hi, [](uint32_t x, uint64_t v) { return uint64_t(x) < v; }) - doc);
		const uint64_t m = b - a;
		int rc;
		if (pos_off) {
			std::vector<uint32_t> po(m + 1);
			for (uint64_t i = 0; i <= m; ++i) po[i] = pos_off[a + i] - pos_off[a];
			rc = rxgpu_ft_set_word_positions(sh, word_id, m, m ? doc + a : nullptr, po.data(), m ? fpos + pos_off[a] : nullptr);
		} else {
			std::vector<uint32_t> eo(m + 1);
			for (uint64_t i = 0; i <= m; ++i) eo[i] = ent_off[a + i] - ent_off[a];
			const uint32_t e0 = m ? ent_off[a] : 0;
			rc = rxgpu_ft_set_word(sh, word_id, m, m ? doc + a : nullptr, eo.data(), ent_field + e0, ent_tf + e0, ent_first_pos + e0);
		}
		if (rc) return rc;
		std::lock_guard<std::mutex> slk(sh->mtx);
		std::unique_lock<std::shared_mutex> dict_lk(sh->dict_mtx);
		sh->words[word_id].df = n;   // (an empty fragment keeps its entry: the word's row exists on every shard)
	}
	return RXGPU_OK;
}

void rxgpu_ft_destroy(rxgpu_ft_index* h) {
	if (!h) return;
	if (h->shard_set) {
		DevGuard dgs(h->device);
		ft_shards_destroy(h->shard_set);
		h->shard_set = nullptr;
	}
	DevGuard dg(h->device);
	(void)rxgpu::device_wait_all(h->device);
	for (auto& kv : h->words) kv.second.release();
	for (void* p : {static_cast<void*>(h->d_words), static_cast<void*>(h->d_avg), static_cast<void*>(h->d_removed), static_cast<void*>(h->d_removed_bits)}) {
		if (p) (void)hipFree(p);
	}
	for (auto& l : h->lanes) release_lane(l.get());
	for (auto& l : h->batch_lanes) release_lane(l.get());
	h->d_batch_plans.release();
	if (h->h_batch_plans) (void)hipHostFree(h->h_batch_plans);
	if (h->ev_ba) (void)hipEventDestroy(h->ev_ba);
	if (h->ev_bb) (void)hipEventDestroy(h->ev_bb);
	if (h->batch_stream) (void)hipStreamDestroy(h->batch_stream);
	release_lane(h);
	delete h;
}

int rxgpu_ft_set_docs(rxgpu_ft_index* h, uint64_t total_docs, const float* words_in_field, const float* avg_words, const uint8_t* removed) {
	RX_CHECK(h && words_in_field && avg_words, RXGPU_ERR_PARAMS, "rxgpu_ft_set_docs: null argument");
	RX_CHECK(total_docs >= 1 && total_docs < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "rxgpu_ft_set_docs: total_docs out of range");
	if (h->shard_set) return ft_shards_set_docs(h, total_docs, words_in_field, avg_words, removed);
	std::lock_guard<std::mutex> lk(h->mtx);
	std::unique_lock<std::shared_mutex> dict_lk(h->dict_mtx);   // no merge on any lane reads the dictionary meanwhile
	DevGuard dg(h->device);
	RX_HIP(hipStreamSynchronize(h->stream));
	if (int rc = upload(h->d_words, words_in_field, total_docs * h->num_fields); rc) return rc;
	if (int rc = upload(h->d_avg, avg_words, h->num_fields); rc) return rc;
	bool any_removed = false;
	if (removed) {
		if (int rc = upload(h->d_removed, removed, total_docs); rc) return rc;
		std::vector<uint32_t> bits((total_docs + 31) / 32, 0u);
		for (uint64_t d = 0; d < total_docs; ++d) {
			if (removed[d]) {
				bits[d >> 5] |= 1u << (d & 31);
				any_removed = true;
			}
		}
		if (any_removed) {
			if (int rc = upload(h->d_removed_bits, bits.data(), bits.size()); rc) return rc;
		}
	} else {
		if (h->d_removed) (void)hipFree(h->d_removed);
		h->d_removed = nullptr;
	}
	if (!any_removed && h->d_removed_bits) {
		(void)hipFree(h->d_removed_bits);
		h->d_removed_bits = nullptr;
	}
	h->h_avg.assign(avg_words, avg_words + h->num_fields);
	h->total_docs = total_docs;
	return RXGPU_OK;
}

int rxgpu_ft_set_word(rxgpu_ft_index* h, uint32_t word_id, uint64_t n, const uint32_t* doc, const uint32_t* ent_off, const uint8_t* ent_field,
					  const uint32_t* ent_tf, const uint32_t* ent_first_pos) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null ft index");
	RX_CHECK(n == 0 || (doc && ent_off && ent_field && ent_tf && ent_first_pos), RXGPU_ERR_PARAMS, "rxgpu_ft_set_word: null argument");
	if (h->shard_set) return ft_shards_set_word(h, word_id, n, doc, ent_off, ent_field, ent_tf, ent_first_pos, nullptr, nullptr);
	std::lock_guard<std::mutex> lk(h->mtx);
	std::unique_lock<std::shared_mutex> dict_lk(h->dict_mtx);   // no merge on any lane reads the dictionary meanwhile
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
	{   // range index over the (ascending) document ids: range k starts at the first posting with doc >= k * kFtRangeDocs; the last entry is n
		RX_CHECK(n < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "rxgpu_ft_set_word: posting list too long");
		const uint32_t n_ranges = uint32_t(doc[n - 1] / rxgpu::kFtRangeDocs) + 2;
		std::vector<uint32_t> ro(n_ranges);
		uint64_t i = 0;
		for (uint32_t k = 0; k < n_ranges; ++k) {
			const uint64_t first_doc = uint64_t(k) * rxgpu::kFtRangeDocs;
			while (i < n && doc[i] < first_doc) {
				RX_CHECK(i == 0 || doc[i] > doc[i - 1], RXGPU_ERR_PARAMS, "rxgpu_ft_set_word: document ids must ascend strictly");
				++i;
			}
			ro[k] = uint32_t(i);
		}
		for (; i < n; ++i) RX_CHECK(i == 0 || doc[i] > doc[i - 1], RXGPU_ERR_PARAMS, "rxgpu_ft_set_word: document ids must ascend strictly");
		if (int rc = upload(w.range_off, ro.data(), ro.size()); rc) return rc;
		w.n_ranges = n_ranges;
	}
	w.n = n;
	w.nent = nent;
	w.last_doc = doc[n - 1];
	return RXGPU_OK;
}

// ---------------------------------------------------------------------------------------------------- one merge = one launch train
namespace {

struct QueryTermIn {
	int32_t op;
	const rxgpu_ft_term_opts* opts;
	uint32_t sub_begin, sub_end;
	int32_t phrase_num = -1;   // FtDslOpts::phraseNum: consecutive terms with the same number >= 0 are one phrase (selecterimpl.h:482-572)
	int32_t distance = 1;      // FtDslOpts::distance (the phrase's terms)
};
// multi-word synonyms of a query (rxgpu_ft_query): their terms are terms[first_term ..] of run_merge's list
struct SynonymsIn {
	uint32_t nsyn = 0, first_term = 0;
	const uint32_t* syn_term_off = nullptr;   // [nsyn + 1], relative to first_term
	const uint32_t* part_syn_off = nullptr;   // [nparts + 1]
	const uint32_t* part_syn = nullptr;
	const uint8_t* suppressed = nullptr;      // per sub-term
};
// a query part (PhraseOrTerm, querymergedata.h:145-176): one plain term or the terms [t_begin, t_end) of one phrase
struct QueryPartIn {
	bool phrase;
	uint32_t t_begin, t_end;
};

// the calculator's IDF per sub-term (bm25.h): totalDocCount = totalNumDocs - 1 ("first doc is always empty"), matchedDocCount = |postings|
double subterm_idf(int bm25_type, uint64_t total_docs, uint64_t n) {
	const double td = double(total_docs - 1), md = double(n);
	if (bm25_type == rxgpu::kFtBm25WordCount) return 0.0;                            // TermCount::GetIDF
	if (bm25_type == rxgpu::kFtBm25Classic) return std::log(td / (md + 1)) + 1;      // Bm25Classic::IDF
	double f = n ? std::log((td - md + 1) / md) / std::log(1 + td) : 0.2;            // Bm25Rx::IDF, saturated at 0.2
	if (f < 0.2) f = 0.2;
	return f;
}

size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// Carves `bytes` out of one growable device buffer; every region starts on a 256-byte boundary
struct Carver {
	size_t off = 0;
	size_t take(size_t bytes) {
		const size_t at = off;
		off = align256(off + bytes);
		return at;
	}
};

// FtDslOpts of a term as the kernels want it
int check_term_opts(const QueryTermIn& qt, uint32_t nf, const char* who, bool& same, bool& all_pos) {
	RX_CHECK(qt.opts->field_boost && qt.opts->need_sum_rank, RXGPU_ERR_PARAMS, std::string(who) + ": null term options");
	uint32_t nsum = 0;
	same = true;
	all_pos = true;
	for (uint32_t f = 0; f < nf; ++f) {
		nsum += qt.opts->need_sum_rank[f] ? 1 : 0;
		same = same && qt.opts->field_boost[f] == qt.opts->field_boost[0];
		all_pos = all_pos && qt.opts->field_boost[f] != 0.0f;
	}
	RX_CHECK(nsum <= 8, RXGPU_ERR_PARAMS, std::string(who) + ": more than 8 fields with needSumRank (GPU engine limit)");
	RX_CHECK(qt.sub_end - qt.sub_begin <= 4096, RXGPU_ERR_PARAMS, std::string(who) + ": more than 4096 sub-terms in one term (GPU engine limit)");
	return RXGPU_OK;
}
void fill_term_cfg(rxgpu::FtTermCfg& tc, const rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const QueryTermIn& qt, bool same, bool all_pos) {
	tc.num_fields = h->num_fields;
	tc.bm25_type = cfg->bm25_type;
	tc.words = h->d_words;
	tc.avg_words = h->d_avg;
	tc.k1 = cfg->bm25_k1;
	tc.b = cfg->bm25_b;
	tc.summation_ratio = cfg->summation_ranks_by_fields_ratio;
	tc.opts_boost = qt.opts->boost;
	tc.term_len_boost_in = qt.opts->term_len_boost;
	tc.op = qt.op;
	tc.same_boost = same ? 1 : 0;
	tc.all_pos_boost = all_pos ? 1 : 0;
}
// the per-field FTConfig parameters as floats (bound() takes float arguments): 6 x nf
void stage_field_cfg(float* fc, const rxgpu_ft_config* cfg, uint32_t nf) {
	for (uint32_t f = 0; f < nf; ++f) {
		fc[0 * nf + f] = float(cfg->bm25_boost[f]);
		fc[1 * nf + f] = float(cfg->bm25_weight[f]);
		fc[2 * nf + f] = float(cfg->term_len_boost[f]);
		fc[3 * nf + f] = float(cfg->term_len_weight[f]);
		fc[4 * nf + f] = float(cfg->position_boost[f]);
		fc[5 * nf + f] = float(cfg->position_weight[f]);
	}
}
void point_term_cfg(rxgpu::FtTermCfg& tc, const float* d_fc, const float* d_field_boost, const uint8_t* d_need_sum, uint32_t nf) {
	tc.field_boost = d_field_boost;
	tc.need_sum_rank = d_need_sum;
	tc.bm25_boost = d_fc + 0 * nf;
	tc.bm25_weight = d_fc + 1 * nf;
	tc.term_len_boost = d_fc + 2 * nf;
	tc.term_len_weight = d_fc + 3 * nf;
	tc.position_boost = d_fc + 4 * nf;
	tc.position_weight = d_fc + 5 * nf;
}
rxgpu::FtPosSubterm word_subterm(const rxgpu_ft_word& w, int bm25_type, uint64_t N, float proc) {
	rxgpu::FtPosSubterm ft{};
	ft.n = w.n;
	ft.doc = w.doc;
	ft.ent_off = w.ent_off;
	ft.ent_field = w.ent_field;
	ft.ent_tf = w.ent_tf;
	ft.ent_first_pos = w.ent_first_pos;
	ft.pos_off = w.pos_off;
	ft.fpos = w.fpos;
	ft.idf = subterm_idf(bm25_type, N, word_df(w));
	ft.proc = proc;
	ft.range_off = w.range_off;
	ft.n_ranges = w.n_ranges;
	return ft;
}

// A phrase between its admission pass and the rest (a document-range shard: the sharded layer settles the admission cut of the WHOLE index —
// at most mergeLimit documents in (row, document) order, phrasemerger.h:341 — before any shard goes on; finish_phrase)
struct PhraseCtx {
	rxgpu::FtPhrasePlan p{};
	std::vector<uint32_t> row_sub, shard_row_sub;
	std::vector<int32_t> shard_row_grid;
	std::vector<uint32_t> row_admitted;   // by the row numbering all shards share: documents this shard admitted for that row
	uint32_t n_rows0 = 0, n_ranges = 0, admitted = 0;
	uint64_t sum_caps = 0;
	size_t phrase_index = 0;
	bool shard = false;
};
// One phrase through ft_phrase.hip: the rows the main merge reads instead of words
struct PhraseRows {
	std::vector<rxgpu::FtPosSubterm> rows;   // non-empty rows, first-term sub-term order; device arrays live in the handle's phrase buffers
	uint32_t admitted = 0;                   // PhraseMerger::NumDocsMerged()
	uint32_t proc16 = 0;                     // PhraseResults::CalcProc16
	uint64_t postings = 0;                   // postings of the phrase's words (statistics)
	std::shared_ptr<PhraseCtx> pending;      // admission ran, finish_phrase has not yet (run_phrase(..., first_half_only))
};
int finish_phrase(rxgpu_ft_index* h, const float* procs, PhraseCtx& c, PhraseRows& out, const char* who);
int run_phrase(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const std::vector<QueryTermIn>& terms, const QueryPartIn& part, const uint32_t* word_ids,
			   const float* procs, const uint8_t* d_excluded, size_t phrase_index, PhraseRows& out, const char* who, bool first_half_only = false) {
	const uint32_t nf = h->num_fields, T = part.t_end - part.t_begin;
	const uint64_t N = h->total_docs;
	const int bm25_type = cfg->bm25_type;
	std::vector<rxgpu::FtPosSubterm> subs;
	std::vector<rxgpu::FtTermCfg> tcfg(T);
	std::vector<int32_t> distance(T);
	std::vector<rxgpu::FtGridEntry> grid;
	std::vector<uint32_t> row_sub;   // first term: position of the row's sub-term in the caller's list
	uint64_t grid_blocks = 0, term0_vdocs = 0, term0_df = 0;
	long long sum_proc = 0;
	// A document-range shard: the rows of the phrase are numbered alike on every shard — one per sub-term of the first term that
	// holds postings ANYWHERE in the index (word_df), with or without postings in this shard's documents — because the sharded layer adds the
	// shards' [rows][ranges] tables up.  shard_row_grid: the row's entry in `grid`, -1 when this shard holds none of its postings.
	const bool shard = h->sh_total > 1;
	std::vector<uint32_t> shard_row_sub;
	std::vector<int32_t> shard_row_grid;
	auto empty_row = [&](uint32_t si) {
		rxgpu::FtPosSubterm row{};
		row.proc = procs[si];
		row.phrase = 1;
		return row;
	};
	for (uint32_t k = 0; k < T; ++k) {
		const QueryTermIn& qt = terms[part.t_begin + k];
		bool same, all_pos;
		if (int rc = check_term_opts(qt, nf, who, same, all_pos); rc) return rc;
		fill_term_cfg(tcfg[k], h, cfg, qt, same, all_pos);
		distance[k] = qt.distance;
		tcfg[k].sub_begin = uint32_t(subs.size());
		if (qt.sub_end > qt.sub_begin) sum_proc = (long long)(float(sum_proc) + procs[qt.sub_begin]);   // CalcProc16: long long += float, term by term
		for (uint32_t si = qt.sub_begin; si < qt.sub_end; ++si) {
			const rxgpu_ft_word& w = h->dict().find(word_ids[si])->second;
			RX_CHECK(w.n == 0 || w.fpos, RXGPU_ERR_LOGIC, std::string(who) + ": the word was uploaded without positions (rxgpu_ft_set_word_positions)");
			RX_CHECK(si == qt.sub_begin || procs[si] <= procs[si - 1], RXGPU_ERR_PARAMS,
					 std::string(who) + ": sub-terms must be sorted by proc, descending (SortSubterms)");
			if (k == 0) {
				term0_vdocs += w.n;
				term0_df += word_df(w);
				if (shard && word_df(w)) {
					shard_row_sub.push_back(si);
					shard_row_grid.push_back(w.n ? int32_t(grid.size()) : -1);
				}
			}
			out.postings += w.n;
			if (!w.n) continue;
			rxgpu::FtPosSubterm ft = word_subterm(w, bm25_type, N, procs[si]);
			ft.term = k;
			ft.ord_in_term = uint16_t(si - qt.sub_begin);
			if (k == 0) {
				grid.push_back({uint32_t(grid_blocks), uint32_t(subs.size())});
				grid_blocks += rxgpu::ft_pass_blocks(w.n);
				row_sub.push_back(si);
			}
			subs.push_back(ft);
		}
		tcfg[k].sub_end = uint32_t(subs.size());
	}
	RX_CHECK(sum_proc >= 0 && sum_proc < 65535, RXGPU_ERR_PARAMS, std::string(who) + ": the procs of a phrase's terms add up to 65535 or more");
	out.proc16 = uint32_t(sum_proc);
	const uint32_t n_rows0 = uint32_t(grid.size());
	(void)term0_df;   // (the admission cut of the whole index — phrasemerger.h:341 — is settled by the sharded layer between the two halves)
	const uint64_t max_merged = std::min<uint64_t>(cfg->merge_limit, term0_vdocs);   // phrasemerger.h:341
	if (!n_rows0 || !max_merged) {   // the first term matched nothing (here): no document (of this shard) holds the phrase
		for (const uint32_t si : shard_row_sub) out.rows.push_back(empty_row(si));
		return RXGPU_OK;
	}
	RX_CHECK(grid_blocks * rxgpu::kFtBlockPostings < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, std::string(who) + ": more than 2^32 (padded) postings in one phrase term");
	const uint32_t n_ranges = uint32_t((N + rxgpu::kFtRangeDocs - 1) / rxgpu::kFtRangeDocs);
	const size_t M = size_t(max_merged);

	if (h->d_phrase_a.size() <= phrase_index) {
		h->d_phrase_a.resize(phrase_index + 1);
		h->d_phrase_b.resize(phrase_index + 1);
	}
	Carver ca;
	const size_t o_subs = ca.take(subs.size() * sizeof(rxgpu::FtPosSubterm));
	const size_t o_terms = ca.take(size_t(T) * sizeof(rxgpu::FtTermCfg));
	const size_t o_dist = ca.take(size_t(T) * 4);
	const size_t o_grid = ca.take(grid.size() * sizeof(rxgpu::FtGridEntry));
	const size_t cfg_floats = size_t(6) * nf + size_t(T) * nf;
	const size_t o_fc = ca.take(cfg_floats * 4 + size_t(T) * nf);
	const size_t plan_bytes = ca.off;
	const size_t o_zero = ca.off;
	const size_t o_lb = ca.take(size_t(grid_blocks) * 8);
	const size_t o_sync = ca.take(8 * 4);
	const size_t zero_bytes = ca.off - o_zero;
	const size_t o_sdoc = ca.take(M * 4), o_srow = ca.take(M * 4), o_scap = ca.take(M * 4), o_sproc = ca.take(M * 4), o_sfield = ca.take(M);
	const size_t o_spos = ca.take(M * 8), o_snpos = ca.take(M * 4);
	rxgpu_devbuf& da = h->d_phrase_a[phrase_index];
	if (int rc = da.ensure(ca.off); rc) return rc;
	char* base = static_cast<char*>(da.ptr);
	if (int rc = h->ensure_pinned(std::max<size_t>(plan_bytes, (size_t(8) + n_rows0) * 4 + 256)); rc) return rc;
	char* hp = static_cast<char*>(h->h_pinned);
	std::memset(hp, 0, plan_bytes);
	float* fc = reinterpret_cast<float*>(hp + o_fc);
	uint8_t* need_sum = reinterpret_cast<uint8_t*>(fc + cfg_floats);
	const float* d_fc = reinterpret_cast<const float*>(base + o_fc);
	const uint8_t* d_need_sum = reinterpret_cast<const uint8_t*>(d_fc + cfg_floats);
	stage_field_cfg(fc, cfg, nf);
	for (uint32_t k = 0; k < T; ++k) {
		const QueryTermIn& qt = terms[part.t_begin + k];
		for (uint32_t f = 0; f < nf; ++f) {
			fc[size_t(6 + k) * nf + f] = qt.opts->field_boost[f];
			need_sum[size_t(k) * nf + f] = qt.opts->need_sum_rank[f];
		}
		point_term_cfg(tcfg[k], d_fc, d_fc + size_t(6 + k) * nf, d_need_sum + size_t(k) * nf, nf);
	}
	std::memcpy(hp + o_subs, subs.data(), subs.size() * sizeof(rxgpu::FtPosSubterm));
	std::memcpy(hp + o_terms, tcfg.data(), tcfg.size() * sizeof(rxgpu::FtTermCfg));
	std::memcpy(hp + o_dist, distance.data(), distance.size() * 4);
	std::memcpy(hp + o_grid, grid.data(), grid.size() * sizeof(rxgpu::FtGridEntry));
	hipStream_t st = h->stream;
	RX_HIP(hipMemcpyAsync(base, hp, plan_bytes, hipMemcpyHostToDevice, st));
	RX_HIP(hipMemsetAsync(base + o_zero, 0, zero_bytes, st));

	rxgpu::FtPhrasePlan p{};
	p.subs = reinterpret_cast<const rxgpu::FtPosSubterm*>(base + o_subs);
	p.terms = reinterpret_cast<const rxgpu::FtTermCfg*>(base + o_terms);
	p.distance = reinterpret_cast<const int32_t*>(base + o_dist);
	p.grid = reinterpret_cast<const rxgpu::FtGridEntry*>(base + o_grid);
	p.nterms = T;
	p.n_grid = n_rows0;
	p.grid_blocks = uint32_t(grid_blocks);
	p.n_rows0 = n_rows0;
	p.max_merged = uint32_t(max_merged);
	p.n_ranges = n_ranges;
	p.total_docs = N;
	p.distance_weight = float(cfg->distance_weight);
	p.distance_boost = float(cfg->distance_boost);
	p.removed = h->d_removed;
	p.excluded = d_excluded;
	p.lookback = reinterpret_cast<unsigned long long*>(base + o_lb);
	p.sync = reinterpret_cast<uint32_t*>(base + o_sync);
	p.slot_doc = reinterpret_cast<uint32_t*>(base + o_sdoc);
	p.slot_row = reinterpret_cast<uint32_t*>(base + o_srow);
	p.slot_cap = reinterpret_cast<uint32_t*>(base + o_scap);
	p.slot_proc = reinterpret_cast<float*>(base + o_sproc);
	p.slot_field = reinterpret_cast<uint8_t*>(base + o_sfield);
	p.slot_pos = reinterpret_cast<uint64_t*>(base + o_spos);
	p.slot_npos = reinterpret_cast<uint32_t*>(base + o_snpos);
	if (!h->ev_pha) {
		RX_HIP(hipEventCreate(&h->ev_pha));
		RX_HIP(hipEventCreate(&h->ev_phb));
	}
	RX_HIP(hipEventRecord(h->ev_pha, st));
	RX_HIP(rxgpu::launch_ft_phrase_admit(p, st));
	RX_HIP(hipMemcpyAsync(hp, p.sync, 8 * 4, hipMemcpyDeviceToHost, st));
	RX_HIP(hipStreamSynchronize(st));
	const uint32_t* sy = reinterpret_cast<const uint32_t*>(hp);
	RX_CHECK(sy[1] == 0, RXGPU_ERR_DEVICE, std::string(who) + ": ordered look-back timed out on the device (phrase admission)");
	const uint32_t admitted = sy[2];
	const uint64_t sum_caps = uint64_t(sy[4]) | (uint64_t(sy[5]) << 32);
	RX_CHECK(admitted <= max_merged, RXGPU_ERR_DEVICE, std::string(who) + ": corrupt phrase admission count");
	RX_CHECK(sum_caps < (1ull << 31), RXGPU_ERR_PARAMS, std::string(who) + ": more than 2^31 positions in the documents of one phrase (GPU engine limit)");
	out.admitted = admitted;
	auto ctx = std::make_shared<PhraseCtx>();
	ctx->p = p;
	ctx->row_sub = row_sub;
	ctx->shard_row_sub = shard_row_sub;
	ctx->shard_row_grid = shard_row_grid;
	ctx->n_rows0 = n_rows0;
	ctx->n_ranges = n_ranges;
	ctx->admitted = admitted;
	ctx->sum_caps = sum_caps;
	ctx->phrase_index = phrase_index;
	ctx->shard = shard;
	if (first_half_only) {   // what this shard admitted, row by row (slot order IS (row, document) order): the sharded layer's cut needs it
		std::vector<uint32_t> slot_row(admitted);
		if (admitted) RX_HIP(hipMemcpy(slot_row.data(), p.slot_row, size_t(admitted) * 4, hipMemcpyDeviceToHost));
		std::vector<uint32_t> by_grid(n_rows0, 0);
		for (const uint32_t r : slot_row) {
			RX_CHECK(r < n_rows0, RXGPU_ERR_DEVICE, std::string(who) + ": corrupt phrase admission rows");
			++by_grid[r];
		}
		ctx->row_admitted.assign(shard_row_sub.size(), 0);
		for (size_t j = 0; j < shard_row_sub.size(); ++j) {
			if (shard_row_grid[j] >= 0) ctx->row_admitted[j] = by_grid[size_t(shard_row_grid[j])];
		}
		out.pending = std::move(ctx);
		return RXGPU_OK;
	}
	return finish_phrase(h, procs, *ctx, out, who);
}

// The rest of a phrase behind its admission: the admitted documents term by term (ft_phrase_docs), the packed rows (ft_phrase_pack).
// c.admitted may have been LOWERED by the sharded layer (the cut of the whole index fell inside or before this shard's documents): the
// admitted documents are a prefix of the slots, so the count on the device is simply overwritten.
int finish_phrase(rxgpu_ft_index* h, const float* procs, PhraseCtx& c, PhraseRows& out, const char* who) {
	rxgpu::FtPhrasePlan& p = c.p;
	const uint32_t admitted = c.admitted, n_rows0 = c.n_rows0, n_ranges = c.n_ranges;
	const uint64_t sum_caps = c.sum_caps;
	const size_t phrase_index = c.phrase_index;
	const bool shard = c.shard;
	const std::vector<uint32_t>&row_sub = c.row_sub, &shard_row_sub = c.shard_row_sub;
	const std::vector<int32_t>& shard_row_grid = c.shard_row_grid;
	hipStream_t st = h->stream;
	char* hp = static_cast<char*>(h->h_pinned);
	auto empty_row = [&](uint32_t si) {
		rxgpu::FtPosSubterm row{};
		row.proc = procs[si];
		row.phrase = 1;
		return row;
	};
	out.admitted = admitted;
	if (shard) RX_HIP(hipMemcpyAsync(p.sync + 2, &c.admitted, 4, hipMemcpyHostToDevice, st));   // (c outlives the copy: the waits below)

	// ---- workspace + packed rows, sized by what the admission found
	const size_t pad = rxgpu::kFtPhraseRowPad;
	const size_t cap_entries = size_t(admitted) + (size_t(n_rows0) + 1) * pad;
	Carver cb;
	const size_t o_ws = cb.take(std::max<uint64_t>(1, 2 * sum_caps) * 8);
	const size_t o_rcnt = cb.take(size_t(n_rows0) * 4), o_rbase = cb.take(size_t(n_rows0) * 4);
	const size_t o_odoc = cb.take(cap_entries * 4), o_orank = cb.take(cap_entries * 4), o_ofield = cb.take(cap_entries), o_opoff = cb.take(cap_entries * 4);
	const size_t o_ofpos = cb.take(std::max<uint64_t>(1, sum_caps) * 8);
	const size_t o_orange = cb.take(size_t(n_rows0) * (n_ranges + 1) * 4);
	const size_t o_hdr = cb.take((size_t(4) + n_rows0) * 4);
	rxgpu_devbuf& db = h->d_phrase_b[phrase_index];
	if (int rc = db.ensure(cb.off); rc) return rc;
	char* bb = static_cast<char*>(db.ptr);
	p.ws = reinterpret_cast<uint64_t*>(bb + o_ws);
	p.row_cnt = reinterpret_cast<uint32_t*>(bb + o_rcnt);
	p.row_base = reinterpret_cast<uint32_t*>(bb + o_rbase);
	p.out_doc = reinterpret_cast<uint32_t*>(bb + o_odoc);
	p.out_rank = reinterpret_cast<float*>(bb + o_orank);
	p.out_field = reinterpret_cast<uint8_t*>(bb + o_ofield);
	p.out_pos_off = reinterpret_cast<uint32_t*>(bb + o_opoff);
	p.out_fpos = reinterpret_cast<uint64_t*>(bb + o_ofpos);
	p.out_range_off = reinterpret_cast<uint32_t*>(bb + o_orange);
	p.out_header = reinterpret_cast<uint32_t*>(bb + o_hdr);
	RX_HIP(rxgpu::launch_ft_phrase_docs(p, admitted, st));
	RX_HIP(rxgpu::launch_ft_phrase_pack(p, st));
	RX_HIP(hipEventRecord(h->ev_phb, st));
	RX_HIP(hipMemcpyAsync(hp, p.out_header, (size_t(4) + n_rows0) * 4, hipMemcpyDeviceToHost, st));
	RX_HIP(hipStreamSynchronize(st));
	float ms = 0.f;
	if (hipEventElapsedTime(&ms, h->ev_pha, h->ev_phb) == hipSuccess) h->stat_ms += ms;
	const uint32_t* hdr = reinterpret_cast<const uint32_t*>(hp);
	RX_CHECK(hdr[0] == admitted && hdr[2] <= admitted, RXGPU_ERR_DEVICE, std::string(who) + ": corrupt phrase header");
	size_t row_base = 0;
	std::vector<rxgpu::FtPosSubterm> packed(n_rows0);   // by grid row; n == 0: the row came out empty
	for (uint32_t r = 0; r < n_rows0; ++r) {
		const uint32_t cnt = hdr[4 + r];
		if (cnt) {
			rxgpu::FtPosSubterm row{};
			row.n = cnt;
			row.doc = p.out_doc + row_base;
			row.pos_off = p.out_pos_off + row_base;
			row.fpos = p.out_fpos;
			row.pre_rank = p.out_rank + row_base;
			row.pre_field = p.out_field + row_base;
			row.proc = procs[row_sub[r]];
			row.range_off = p.out_range_off + size_t(r) * (n_ranges + 1);
			row.n_ranges = n_ranges;
			row.phrase = 1;
			packed[r] = row;
			if (!shard) out.rows.push_back(row);
		}
		row_base += (size_t(cnt) + 1 + pad - 1) / pad * pad;
	}
	for (size_t j = 0; j < shard_row_sub.size(); ++j) {   // a shard: every row of the index, the empty ones included
		const int32_t g = shard_row_grid[j];
		out.rows.push_back(g >= 0 && packed[size_t(g)].n ? packed[size_t(g)] : empty_row(shard_row_sub[j]));
	}
	return RXGPU_OK;
}

constexpr std::chrono::milliseconds kResidentPatience{2000};
// what THIS thread believes about its resident session (one text index at a time per thread: HybridQueryResident runs its steps in a row)
thread_local const rxgpu_ft_index* tl_res_handle = nullptr;
thread_local uint64_t tl_res_generation = 0;

// `lk` holds h->mtx.  Waits until no OTHER thread's session is open (bounded), then opens one for the caller.
void open_resident_session(rxgpu_ft_index* h, std::unique_lock<std::mutex>& lk) {
	const auto me = std::this_thread::get_id();
	if (h->res_session && h->res_owner != me) {
		(void)h->res_cv.wait_for(lk, kResidentPatience, [&] { return !h->res_session; });   // timed out: the session is taken over below
	}
	h->res_session = true;
	h->res_owner = me;
	h->res_generation += 1;
	tl_res_handle = h;
	tl_res_generation = h->res_generation;
}
// `lk` holds h->mtx.  RXGPU_OK when the caller may use the lane for the prepare / fuse step: it owns the open session, or it never opened one
// (a query whose FT side merged nothing) and nobody else's is open (waited for, bounded).
int check_resident_session(rxgpu_ft_index* h, std::unique_lock<std::mutex>& lk, const char* who) {
	const auto me = std::this_thread::get_id();
	if (tl_res_handle == h && tl_res_generation != 0) {
		if (!(h->res_session && h->res_owner == me && h->res_generation == tl_res_generation)) {
			tl_res_generation = 0;
			set_error(std::string(who) + ": this thread's resident merge was replaced by another caller's (its session was not fused within 2 s)");
			return RXGPU_ERR_LOGIC;
		}
		return RXGPU_OK;
	}
	if (h->res_session && h->res_owner != me) {
		if (!h->res_cv.wait_for(lk, kResidentPatience, [&] { return !h->res_session; })) {
			set_error(std::string(who) + ": another caller's resident merge is parked on this index");
			return RXGPU_ERR_LOGIC;
		}
	}
	// a fusion without a resident merge in front (the query's FT side merged nothing): a session of its own with an empty FT side, so that
	// nobody else's prepare lands between this caller's prepare and its fuse
	if (!h->res_session) {
		h->res_pending = false;
		h->res_cap = 0;
		h->prep_done = false;
	}
	h->res_session = true;
	h->res_owner = me;
	h->res_generation += 1;
	tl_res_handle = h;
	tl_res_generation = h->res_generation;
	return RXGPU_OK;
}
void close_resident_session(rxgpu_ft_index* h) {
	if (h->res_session && h->res_owner == std::this_thread::get_id()) {
		h->res_session = false;
		h->res_cv.notify_all();
	}
	if (tl_res_handle == h) tl_res_generation = 0;
}

// A resident merge was enqueued and nobody looked at its header yet: wait for it, check the look-back word, settle the kept-clean state.
int finish_pending(rxgpu_ft_index* h, const char* who) {
	if (!h->res_pending) return RXGPU_OK;
	h->res_pending = false;
	RX_HIP(hipStreamSynchronize(h->stream));
	uint32_t hdr[4] = {0, 0, 0, 0};
	RX_HIP(hipMemcpy(hdr, h->d_out.ptr, sizeof(hdr), hipMemcpyDeviceToHost));
	float ms = 0.f;
	if (h->ev_a && hipEventElapsedTime(&ms, h->ev_a, h->ev_b) == hipSuccess) h->stat_ms += ms;
	RX_CHECK(hdr[1] == 0, RXGPU_ERR_DEVICE, std::string(who) + ": ordered look-back timed out on the device");
	h->clean_dirty = false;
	return RXGPU_OK;
}

// MergeDataAreas<Area>: what the caller wants back besides the merged documents (rxgpu_ft_merge_query_areas_raw)
struct AreasOut {
	uint32_t max_areas = 0;      // FTConfig::maxAreasInDoc
	uint32_t* cnt = nullptr;     // [cap][num_fields]
	uint32_t* areas = nullptr;   // [cap][num_fields][max_areas][3]
};

// One merge between the building of its plan and the unpacking of its result.
struct MergeJob {
	rxgpu::FtPlan p{};
	const rxgpu::FtPlan* d_plan = nullptr;   // the plan where the kernels read it (HBM, behind the rest of the plan)
	void* dev_base = nullptr;                // where the staged plan goes (the lane's state buffer)
	uint64_t max_merged = 0, merged_postings = 0;
	size_t plan_bytes = 0;
	void* hp_dev = nullptr;                  // the lane's pinned staging buffer as the device sees it
	uint32_t nsyn = 0;
	bool empty = false;                      // min(mergeLimit, totalORVids) == 0: nothing is merged
	size_t area_hdr_bytes = 0, area_bytes = 0;   // MergeDataAreas: the two regions of the lane's d_areas
};

// Which launch train runs a merge: -1 the host decides per query (ft_sparse_eligible + a density test), 0 always the dense train
// (ft_merge.hip), 1 the sparse train (ft_sparse.hip) whenever the query is eligible.  RXGPU_FT_TRAIN=dense|sparse presets it, read once;
// rxgpu_ft_set_train_mode changes it (tests, benchmarks).
std::atomic<int>* ft_train_mode() {
	static std::atomic<int> mode{[] {
		const char* e = std::getenv("RXGPU_FT_TRAIN");
		if (e && std::strcmp(e, "dense") == 0) return 0;
		if (e && std::strcmp(e, "sparse") == 0) return 1;
		return -1;
	}()};
	return &mode;
}

// The sparse train (ft_sparse.hip) derives every per-document fact from one bitmap per sub-term and ranks a document only once its merge slot
// is known.  That is the reference's merge exactly when
//   * the query is made of plain terms (no phrase rows, no multi-word synonyms, no areas) with at most kFtSparseSubs sub-terms,
//   * every field of every merged term has the same positive boost — calcTermBitmask / calcTermScores then never look at an occurrence's
//     fields (mergerimpl.h:252-324: allFieldsHaveSameBoost; checkFieldsRelevance is true for every occurrence),
//   * calcTermRank cannot return 0 for any posting, so that "added by its first posting with a non-zero rank" (merger.h:161-180) is "added
//     by its first posting": Bm25Rx / TermCount (positive, finite for avg_words > 0), every weight below 1 and every boost >= 0, which
//     bounds each factor of phrasemergerimpl.h:51-63 from below by (1 - weight) > 0; the product's lower bound must stay a normal float.
bool ft_sparse_eligible(const rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const std::vector<QueryTermIn>& terms, const float* procs, size_t n_subs,
						size_t n_phrases, uint32_t nsyn, uint32_t max_areas) {
	if (h->sh_total > 1 || n_phrases || nsyn || max_areas) return false;
	if (n_subs == 0 || n_subs > rxgpu::kFtSparseSubs || terms.size() > 32) return false;
	if (cfg->bm25_type == rxgpu::kFtBm25Classic) return false;   // TF = count / wordsInDoc: a field without words makes the rank NaN, which is never admitted
	if (!(cfg->bm25_k1 >= 0.0) || !(cfg->bm25_b >= 0.0 && cfg->bm25_b <= 1.0) || !(cfg->summation_ranks_by_fields_ratio >= 0.0)) return false;
	const uint32_t nf = h->num_fields;
	if (h->h_avg.size() != nf) return false;
	double floor_fields = 1.0;   // lower bound of norm * termLenBoost * positionRank over the fields
	for (uint32_t f = 0; f < nf; ++f) {
		if (!(h->h_avg[f] > 0.0f) || !std::isfinite(h->h_avg[f])) return false;
		const double w[3] = {cfg->bm25_weight[f], cfg->term_len_weight[f], cfg->position_weight[f]};
		const double b[3] = {cfg->bm25_boost[f], cfg->term_len_boost[f], cfg->position_boost[f]};
		double fl = 1.0;
		for (int k = 0; k < 3; ++k) {
			if (!(w[k] >= 0.0 && w[k] <= 0.999) || !(b[k] >= 0.0) || !std::isfinite(b[k])) return false;
			fl *= 1.0 - w[k];
		}
		floor_fields = std::min(floor_fields, fl);
	}
	for (const QueryTermIn& qt : terms) {
		if (qt.phrase_num >= 0) return false;
		if (qt.op == 3) continue;   // a NOT term only clears mask bits, whatever its options (excludeTermFromBitmask, mergerimpl.h:276-287)
		const float fb = qt.opts->field_boost[0];
		if (!(fb > 0.0f) || !std::isfinite(fb)) return false;
		for (uint32_t f = 1; f < nf; ++f) {
			if (qt.opts->field_boost[f] != fb) return false;
		}
		if (!(qt.opts->boost > 0.0f) || !std::isfinite(qt.opts->boost) || !(qt.opts->term_len_boost >= 0.0f) || !std::isfinite(qt.opts->term_len_boost)) return false;
		for (uint32_t si = qt.sub_begin; si < qt.sub_end; ++si) {
			if (!(procs[si] > 0.0f) || !std::isfinite(procs[si])) return false;
			if (double(fb) * floor_fields * double(qt.opts->boost) * double(procs[si]) < 1e-30) return false;
		}
	}
	return true;
}

// First half of a merge: the plan (sub-terms, per-part configuration, posting-side grid), the lane's scratch, the plan staged in the lane's
// pinned buffer — FtPlan included, behind the tables it points into — and (import_now) the copy kernel that takes it to HBM.  Everything is
// enqueued on `st`: the lane's own stream for a single merge, the batch stream when Q lanes' merges go into one train.
int prepare_merge(rxgpu_ft_index* h, hipStream_t st, const rxgpu_ft_config* cfg, bool simple, const std::vector<QueryTermIn>& terms, const uint32_t* word_ids,
				  const float* procs, const uint8_t* excluded, bool have_outs, uint64_t cap, const char* who, bool resident, const SynonymsIn* synonyms,
				  MergeJob& job, bool import_now, uint32_t max_areas = 0, std::vector<PhraseRows>* shard_phrases = nullptr, int phrase_mode = 0) {
	// shard_phrases / phrase_mode (document-range shards): 1 = run the query's phrases only and hand their rows out (nothing else is prepared);
	// 2 = the rows come in, `admitted` holding the sum over the shards (the 2-phase estimate is a fact of the whole index).
	using clk = std::chrono::steady_clock;
	const auto t_begin = clk::now();
	auto since = [](clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); };
	const uint32_t nf = h->num_fields;
	const uint64_t N = h->total_docs;
	const uint32_t nterms = uint32_t(terms.size());   // the query parts' terms, then the synonyms' terms
	const uint32_t nsyn = synonyms ? synonyms->nsyn : 0;
	const uint32_t npart_terms = nsyn ? synonyms->first_term : nterms;
	const uint32_t nsyn_terms = nterms - npart_terms;
	const int bm25_type = cfg->bm25_type;
	RX_CHECK(bm25_type >= 0 && bm25_type <= 2, RXGPU_ERR_PARAMS, std::string(who) + ": bm25_type must be 0 (rx), 1 (classic) or 2 (wordCount)");

	// ---- the query parts (selecterimpl.h:482-572): consecutive terms with the same phraseNum >= 0 are one phrase
	std::vector<QueryPartIn> parts;
	for (uint32_t t = 0; t < npart_terms;) {
		if (terms[t].phrase_num < 0) {
			parts.push_back({false, t, t + 1});
			++t;
			continue;
		}
		uint32_t e = t + 1;
		while (e < npart_terms && terms[e].phrase_num == terms[t].phrase_num) ++e;
		parts.push_back({true, t, e});
		t = e;
	}
	const uint32_t nparts = uint32_t(parts.size());
	RX_CHECK(nparts < 0x7FFF, RXGPU_ERR_PARAMS, std::string(who) + ": too many query parts");
	RX_CHECK(!simple || (nparts == 1 && !parts[0].phrase && !nsyn), RXGPU_ERR_LOGIC, std::string(who) + ": a phrase is not a Simple() query");
	if (nsyn) {
		RX_CHECK(synonyms->syn_term_off && synonyms->part_syn_off && synonyms->syn_term_off[0] == 0 && synonyms->syn_term_off[nsyn] == nsyn_terms &&
					 synonyms->part_syn_off[0] == 0,
				 RXGPU_ERR_PARAMS, std::string(who) + ": inconsistent synonym tables");
		for (uint32_t k = 0; k < synonyms->part_syn_off[nparts]; ++k) {
			RX_CHECK(synonyms->part_syn && synonyms->part_syn[k] < nsyn, RXGPU_ERR_PARAMS, std::string(who) + ": synonym id out of range");
		}
	}

	// ---- the plan: sub-terms, per-part configuration, the posting-side grid
	std::vector<rxgpu::FtPosSubterm> subs;
	std::vector<rxgpu::FtTermCfg> tcfg(nparts + nsyn_terms);
	std::vector<rxgpu::FtGridEntry> merge_grid;
	std::vector<uint64_t> term_postings(nterms, 0);
	uint64_t total_vids = 0, merged_postings = 0, merge_blocks = 0;
	for (uint32_t t = 0; t < nterms; ++t) {
		const QueryTermIn& qt = terms[t];
		for (uint32_t si = qt.sub_begin; si < qt.sub_end; ++si) {
			auto it = h->dict().find(word_ids[si]);
			RX_CHECK(it != h->dict().end(), RXGPU_ERR_NOTFOUND, std::string(who) + ": unknown word id");
			// the kernels index words_in_field[doc * fields + f] and an (N + 31) / 32-word mask by document: a list reaching past the
			// documents rxgpu_ft_set_docs described would read and write out of bounds (set_docs may follow the words, so it is checked here)
			RX_CHECK(it->second.n == 0 || it->second.last_doc < N, RXGPU_ERR_PARAMS,
					 std::string(who) + ": a posting list holds a document id >= total_docs (rxgpu_ft_set_docs)");
			term_postings[t] += word_df(it->second);   // (a document-range shard: the whole index's count — limits and gates are global facts)
		}
		total_vids += term_postings[t];   // totalORVids: MaxVDocs of every term, whatever its operator and inside phrases too (selecterimpl.h:546)
	}
	RX_CHECK(total_vids < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, std::string(who) + ": more than 2^32 postings in one merge");
	const uint64_t max_merged = std::min<uint64_t>(cfg->merge_limit, total_vids);   // Merge(): min(mergeLimit, totalORVids)
	if (max_merged == 0) {
		job.empty = true;
		return RXGPU_OK;
	}
	RX_CHECK(resident || (cap >= max_merged && have_outs), RXGPU_ERR_OVERFLOW, std::string(who) + ": output buffers too small");

	bool any_phrase = false;
	for (const QueryPartIn& part : parts) any_phrase = any_phrase || part.phrase;
	// the launch train: the sparse one for eligible queries whose postings lie on a fraction of the documents (or on request)
	bool sparse = false;
	{
		const int mode = ft_train_mode()->load(std::memory_order_relaxed);
		size_t n_subs_all = 0;
		uint64_t local_postings = 0;
		for (uint32_t t = 0; t < nterms; ++t) {
			for (uint32_t si = terms[t].sub_begin; si < terms[t].sub_end; ++si) {
				const rxgpu_ft_word& w = h->dict().find(word_ids[si])->second;
				if (!word_df(w)) continue;
				++n_subs_all;
				local_postings += w.n;
			}
		}
		if (mode != 0 && ft_sparse_eligible(h, cfg, terms, procs, n_subs_all, any_phrase ? 1 : 0, nsyn, max_areas)) {
			sparse = mode == 1 || local_postings * 10 <= N * 3;   // dense queries keep the dense train (posting-parallel ranking, per-range workgroups)
		}
	}
	const uint8_t* d_excluded = nullptr;
	const uint32_t* d_excluded_bits = nullptr;
	if (excluded && sparse) {   // the sparse train reads docsExcluded as one bit per document
		std::vector<uint32_t> bits((N + 31) / 32, 0u);
		for (uint64_t d = 0; d < N; ++d) bits[d >> 5] |= (excluded[d] ? 1u : 0u) << (d & 31);
		if (int rc = h->d_excl.ensure(bits.size() * 4); rc) return rc;
		RX_HIP(hipMemcpyAsync(h->d_excl.ptr, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, st));   // (pageable source: the copy is staged before the call returns)
		d_excluded_bits = static_cast<const uint32_t*>(h->d_excl.ptr);
	} else if (excluded) {
		if (int rc = h->d_excl.ensure(N); rc) return rc;
		RX_HIP(hipMemcpyAsync(h->d_excl.ptr, excluded, N, hipMemcpyHostToDevice, st));
		d_excluded = static_cast<const uint8_t*>(h->d_excl.ptr);
	}
	// ---- phrases first (Merger::init, merger.h:73-81): every PhraseMerger runs before the query parts are looked at
	std::vector<PhraseRows> phrase_rows(nparts);
	size_t n_phrases = 0;
	if (phrase_mode == 2) {
		RX_CHECK(shard_phrases && shard_phrases->size() == nparts, RXGPU_ERR_LOGIC, std::string(who) + ": phrase rows of another query");
		phrase_rows = *shard_phrases;
		for (uint32_t pi = 0; pi < nparts; ++pi) n_phrases += parts[pi].phrase ? 1 : 0;
	} else {
		for (uint32_t pi = 0; pi < nparts; ++pi) {
			if (!parts[pi].phrase) continue;
			if (int rc = run_phrase(h, cfg, terms, parts[pi], word_ids, procs, d_excluded, n_phrases++, phrase_rows[pi], who, phrase_mode == 1); rc) return rc;
		}
		if (phrase_mode == 1) {
			*shard_phrases = std::move(phrase_rows);
			return RXGPU_OK;
		}
	}

	// 2-phase gate, host half (estimateNumDocsInMerge, merger.h:239-267; mergerimpl.h:486-490)
	uint64_t est_or = 0, est_and = UINT64_MAX;
	uint32_t query_len = 0;
	for (uint32_t pi = 0; pi < nparts; ++pi) {
		const QueryPartIn& part = parts[pi];
		query_len += part.t_end - part.t_begin;
		const int32_t op = terms[part.t_begin].op;   // PhraseResults::Op(): its first term's
		if (op == 3) continue;
		uint64_t num_docs = part.phrase ? phrase_rows[pi].admitted : term_postings[part.t_begin];
		if (nsyn) {   // + the first term of every synonym of the part (merger.h:251-255)
			for (uint32_t k = synonyms->part_syn_off[pi]; k < synonyms->part_syn_off[pi + 1]; ++k) {
				const uint32_t sy = synonyms->part_syn[k];
				if (synonyms->syn_term_off[sy + 1] > synonyms->syn_term_off[sy]) num_docs += term_postings[npart_terms + synonyms->syn_term_off[sy]];
			}
		}
		if (op == 2) {
			est_and = std::min(est_and, num_docs);
		} else {
			est_or += num_docs;
		}
	}
	const bool prescore = !simple && std::min(std::min(est_or, est_and), N) > cfg->merge_limit && N > cfg->merge_limit;

	uint16_t qp = 0, last_term_qp = 0;
	for (uint32_t pi = 0; pi < nparts; ++pi) {
		const QueryPartIn& part = parts[pi];
		const QueryTermIn& qt = terms[part.t_begin];
		rxgpu::FtTermCfg& tc = tcfg[pi];
		if (part.phrase) {
			// the phrase as one part: its rows carry rank and field, every document counts for the masks, the pre-score adds CalcProc16
			fill_term_cfg(tc, h, cfg, qt, true, true);
			tc.opts_boost = 1.0f;
			tc.phrase = 1;
			tc.phrase_proc16 = phrase_rows[pi].proc16;
			tc.sub_begin = uint32_t(subs.size());
			if (qt.op != 3) ++qp;
			for (rxgpu::FtPosSubterm ft : phrase_rows[pi].rows) {
				ft.term = pi;
				ft.qp = qt.op == 3 ? 0 : qp;
				ft.prev_term_qp = last_term_qp;
				ft.ord_in_term = uint16_t(subs.size() - tc.sub_begin);
				const uint32_t sub_index = uint32_t(subs.size());
				if (qt.op != 3) {
					ft.row = uint32_t(merge_grid.size());
					merge_grid.push_back({uint32_t(merge_blocks), sub_index});
					merge_blocks += rxgpu::ft_pass_blocks(ft.n);
					merged_postings += ft.n;
				}
				subs.push_back(ft);
			}
			tc.sub_end = uint32_t(subs.size());
			h->stat_postings += phrase_rows[pi].postings;
			continue;
		}
		bool same, all_pos;
		if (int rc = check_term_opts(qt, nf, who, same, all_pos); rc) return rc;
		fill_term_cfg(tc, h, cfg, qt, same, all_pos);
		tc.sub_begin = uint32_t(subs.size());
		if (qt.op != 3) last_term_qp = ++qp;
		for (uint32_t si = qt.sub_begin; si < qt.sub_end; ++si) {
			const rxgpu_ft_word& w = h->dict().find(word_ids[si])->second;
			RX_CHECK(simple || w.n == 0 || w.fpos, RXGPU_ERR_LOGIC, std::string(who) + ": the word was uploaded without positions (rxgpu_ft_set_word_positions)");
			RX_CHECK(si == qt.sub_begin || procs[si] <= procs[si - 1], RXGPU_ERR_PARAMS,
					 std::string(who) + ": sub-terms must be sorted by proc, descending (SortSubterms)");
			if (!word_df(w)) continue;   // (a shard keeps the row of a word it holds no posting of: rows are numbered alike on every shard)
			rxgpu::FtPosSubterm ft = word_subterm(w, bm25_type, N, procs[si]);
			ft.term = pi;
			ft.qp = qt.op == 3 ? 0 : qp;
			ft.ord_in_term = uint16_t(si - qt.sub_begin);
			ft.row = 0;
			const uint32_t blocks = rxgpu::ft_pass_blocks(w.n);
			const uint32_t sub_index = uint32_t(subs.size());
			if (qt.op != 3) {
				ft.row = uint32_t(merge_grid.size());
				merge_grid.push_back({uint32_t(merge_blocks), sub_index});
				merge_blocks += blocks;
				merged_postings += w.n;
			}
			subs.push_back(ft);
		}
		tc.sub_end = uint32_t(subs.size());
	}
	// ---- the multi-word synonyms' terms behind the parts (mergerimpl.h:509-514): plain mergeTerm calls, every term takes a qp
	const uint32_t n_part_qp = qp;
	std::vector<rxgpu::FtSynonym> syns(nsyn);
	for (uint32_t sy = 0; sy < nsyn; ++sy) {
		syns[sy].term_begin = nparts + synonyms->syn_term_off[sy];
		syns[sy].term_end = nparts + synonyms->syn_term_off[sy + 1];
		syns[sy].nterms = synonyms->syn_term_off[sy + 1] - synonyms->syn_term_off[sy];
		for (uint32_t k = synonyms->syn_term_off[sy]; k < synonyms->syn_term_off[sy + 1]; ++k) {
			const QueryTermIn& qt = terms[npart_terms + k];
			rxgpu::FtTermCfg& tc = tcfg[nparts + k];
			bool same, all_pos;
			if (int rc = check_term_opts(qt, nf, who, same, all_pos); rc) return rc;
			fill_term_cfg(tc, h, cfg, qt, same, all_pos);
			tc.op = 1;   // for ft_ranges: scored like any term (calcTermScores, mergerimpl.h:393-397), never a restriction of its own
			tc.sub_begin = uint32_t(subs.size());
			RX_CHECK(qp < 0x7FFE, RXGPU_ERR_PARAMS, std::string(who) + ": too many query terms");
			++qp;
			for (uint32_t si = qt.sub_begin; si < qt.sub_end; ++si) {
				const rxgpu_ft_word& w = h->dict().find(word_ids[si])->second;
				RX_CHECK(w.n == 0 || w.fpos, RXGPU_ERR_LOGIC, std::string(who) + ": the word was uploaded without positions (rxgpu_ft_set_word_positions)");
				RX_CHECK(si == qt.sub_begin || procs[si] <= procs[si - 1], RXGPU_ERR_PARAMS,
						 std::string(who) + ": sub-terms must be sorted by proc, descending (SortSubterms)");
				if (!word_df(w)) continue;
				rxgpu::FtPosSubterm ft = word_subterm(w, bm25_type, N, procs[si]);
				ft.term = nparts + k;
				ft.qp = qt.op == 3 ? 0 : qp;
				ft.ord_in_term = uint16_t(si - qt.sub_begin);
				ft.suppressed = synonyms->suppressed && synonyms->suppressed[si] ? 1 : 0;
				const uint32_t sub_index = uint32_t(subs.size());
				if (qt.op != 3) {   // mergeTerm returns at once for a NOT term (mergerimpl.h:110-112)
					ft.row = uint32_t(merge_grid.size());
					merge_grid.push_back({uint32_t(merge_blocks), sub_index});
					merge_blocks += rxgpu::ft_pass_blocks(w.n);
					merged_postings += w.n;
				}
				subs.push_back(ft);
			}
			tc.sub_end = uint32_t(subs.size());
		}
		syns[sy].end_qp = qp;
	}
	// the AND parts whose term mask takes their synonyms' masks in (ft_syn_masks)
	std::vector<rxgpu::FtSynMaskJob> syn_jobs;
	std::vector<uint32_t> job_syns, job_part;
	for (uint32_t pi = 0; pi < nparts && nsyn; ++pi) {
		if (terms[parts[pi].t_begin].op != 2 || synonyms->part_syn_off[pi + 1] == synonyms->part_syn_off[pi]) continue;
		rxgpu::FtSynMaskJob job{};
		job.syn_begin = uint32_t(job_syns.size());
		for (uint32_t k = synonyms->part_syn_off[pi]; k < synonyms->part_syn_off[pi + 1]; ++k) job_syns.push_back(synonyms->part_syn[k]);
		job.syn_end = uint32_t(job_syns.size());
		syn_jobs.push_back(job);
		job_part.push_back(pi);
	}
	RX_CHECK(merge_blocks * rxgpu::kFtBlockPostings < 0xFFFFFFFFull, RXGPU_ERR_PARAMS,
			 std::string(who) + ": more than 2^32 (padded) postings in one merge");
	const uint32_t n_rows = uint32_t(merge_grid.size());
	RX_CHECK(n_rows < 0xFFFFu, RXGPU_ERR_PARAMS, std::string(who) + ": more than 65534 merged sub-terms in one query (GPU engine limit)");
	const uint64_t nwords = (N + 31) / 32;
	const size_t M = size_t(max_merged);

	if (max_areas) {   // MergeDataAreas<Area>: plain terms only (a phrase's areas come out of the PhraseMerger's position chains, phrasemerger.h:147-181)
		RX_CHECK(!resident && !nsyn, RXGPU_ERR_LOGIC, std::string(who) + ": areas are built for queries of plain terms (no multi-word synonyms, no resident form)");
		for (const QueryPartIn& part : parts) RX_CHECK(!part.phrase, RXGPU_ERR_LOGIC, std::string(who) + ": a phrase's areas stay on the CPU merger");
		for (const rxgpu::FtPosSubterm& ft : subs) {
			RX_CHECK(ft.n == 0 || ft.fpos, RXGPU_ERR_LOGIC, std::string(who) + ": areas need the words' positions (rxgpu_ft_set_word_positions)");
		}
	}
	h->trace_us[0] += since(t_begin);
	const auto t_stage = clk::now();
	// ---- device scratch (one buffer each for the state and for the packed result)
	Carver cv;
	const size_t o_plan_subs = cv.take(std::max<size_t>(1, subs.size()) * sizeof(rxgpu::FtPosSubterm));
	const uint32_t nplan_terms = nparts + nsyn_terms;
	const size_t o_plan_terms = cv.take(size_t(nplan_terms) * sizeof(rxgpu::FtTermCfg));
	const size_t o_plan_mgrid = cv.take(std::max<size_t>(1, merge_grid.size()) * sizeof(rxgpu::FtGridEntry));
	const size_t cfg_floats = size_t(6) * nf + size_t(nplan_terms) * nf;
	const size_t o_plan_fc = cv.take(cfg_floats * sizeof(float) + size_t(nplan_terms) * nf);
	const size_t o_plan_syns = cv.take(std::max<size_t>(1, syns.size()) * sizeof(rxgpu::FtSynonym));
	const size_t o_plan_jobs = cv.take(std::max<size_t>(1, syn_jobs.size()) * sizeof(rxgpu::FtSynMaskJob));
	const size_t o_plan_jsyn = cv.take(std::max<size_t>(1, job_syns.size()) * 4);
	const size_t o_plan_self = cv.take(sizeof(rxgpu::FtPlan));   // the FtPlan itself: the kernels read it from HBM (a batch of one)
	const size_t plan_bytes = cv.off;   // everything above is uploaded in one copy
	// (a sparse merge keeps nothing per document or per posting in HBM: ft_sparse.hip)
	const size_t o_mask = cv.take(sparse ? 0 : nwords * 4);
	const size_t o_synmask = cv.take(syn_jobs.size() * nwords * 4);
	const size_t o_score = cv.take(prescore && !sparse ? nwords * 32 * 2 : 0);   // padded to whole mask words (ft_preselect_apply reads 32 scores at a time)
	const uint32_t n_ranges = uint32_t((N + rxgpu::kFtRangeDocs - 1) / rxgpu::kFtRangeDocs);
	const size_t o_brec = cv.take(sparse ? 0 : size_t(merged_postings) * sizeof(uint4));
	const size_t o_boff = cv.take(size_t(n_ranges) * 4);
	const size_t o_adders = cv.take(std::max<size_t>(1, size_t(n_rows) * n_ranges) * 4);
	const size_t o_eidx = cv.take(sparse ? 0 : size_t(n_rows) * M * 4);
	const size_t o_efield = cv.take(sparse ? 0 : size_t(n_rows) * M);
	const size_t o_tdoc = cv.take(sparse ? M * 4 : 0);
	const size_t o_tpos = cv.take(sparse ? M * 4 : 0);
	const size_t o_tidx = cv.take(sparse ? M * std::max<size_t>(1, n_rows) * 4 : 0);
	if (int rc = h->d_state.ensure(cv.off); rc) return rc;
	char* base = static_cast<char*>(h->d_state.ptr);
	// the kept-clean tables: sized by the corpus only, so that they stay where they are from merge to merge
	Carver cc;
	const size_t o_hist = cc.take(size_t(rxgpu::kFtHistCopies) * rxgpu::kFtHistStride * 4);   // copies of (fine + coarse)
	const size_t o_lb_pre = cc.take(((nwords + 1023) / 1024) * 8);
	const size_t o_bcnt = cc.take(size_t(n_ranges) * 4);
	const size_t o_sync = cc.take(rxgpu::kFtSyncWords * 4);
	const size_t o_dbg = cc.take(64 * 8);
	const size_t o_lb_units = cc.take(size_t(n_ranges) * 8);
	const size_t o_erank = cc.take(sparse ? 0 : size_t(n_rows) * M * 4);   // last: the regions before it never move when a query needs more rows
	if (h->clean_docs != N || h->d_clean.bytes < cc.off) {
		if (int rc = h->d_clean.ensure(cc.off); rc) return rc;
		h->clean_docs = N;
		h->clean_dirty = true;
	}
	char* cbase = static_cast<char*>(h->d_clean.ptr);
	const size_t out_need = align256(16) + align256(M * 4) * 2 + align256(M * 2) + align256(M);
	if (int rc = h->d_out.ensure(out_need); rc) return rc;
	char* ob = static_cast<char*>(h->d_out.ptr);

	// ---- host staging of the plan (pinned), one upload
	if (int rc = h->ensure_pinned(std::max(plan_bytes, out_need)); rc) return rc;
	char* hp = static_cast<char*>(h->h_pinned);
	std::memset(hp, 0, plan_bytes);
	float* fc = reinterpret_cast<float*>(hp + o_plan_fc);
	const float* d_fc = reinterpret_cast<const float*>(base + o_plan_fc);
	const uint8_t* d_need_sum = reinterpret_cast<const uint8_t*>(d_fc + cfg_floats);
	uint8_t* need_sum = reinterpret_cast<uint8_t*>(fc + cfg_floats);
	stage_field_cfg(fc, cfg, nf);
	for (uint32_t pi = 0; pi < nparts; ++pi) {
		const QueryTermIn& qt = terms[parts[pi].t_begin];
		for (uint32_t f = 0; f < nf; ++f) {   // a phrase part: ones (its rows are ranked already; ft_ranges reads field_boost[0] > 0)
			fc[size_t(6 + pi) * nf + f] = parts[pi].phrase ? 1.0f : qt.opts->field_boost[f];
			need_sum[size_t(pi) * nf + f] = parts[pi].phrase ? uint8_t(0) : qt.opts->need_sum_rank[f];
		}
		point_term_cfg(tcfg[pi], d_fc, d_fc + size_t(6 + pi) * nf, d_need_sum + size_t(pi) * nf, nf);
	}
	for (uint32_t k = 0; k < nsyn_terms; ++k) {
		const QueryTermIn& qt = terms[npart_terms + k];
		const uint32_t ti = nparts + k;
		for (uint32_t f = 0; f < nf; ++f) {
			fc[size_t(6 + ti) * nf + f] = qt.opts->field_boost[f];
			need_sum[size_t(ti) * nf + f] = qt.opts->need_sum_rank[f];
		}
		point_term_cfg(tcfg[ti], d_fc, d_fc + size_t(6 + ti) * nf, d_need_sum + size_t(ti) * nf, nf);
	}
	for (size_t j = 0; j < syn_jobs.size(); ++j) {
		syn_jobs[j].out = reinterpret_cast<uint32_t*>(base + o_synmask) + j * nwords;
		tcfg[job_part[j]].syn_mask = syn_jobs[j].out;
	}
	if (!syns.empty()) std::memcpy(hp + o_plan_syns, syns.data(), syns.size() * sizeof(rxgpu::FtSynonym));
	if (!syn_jobs.empty()) std::memcpy(hp + o_plan_jobs, syn_jobs.data(), syn_jobs.size() * sizeof(rxgpu::FtSynMaskJob));
	if (!job_syns.empty()) std::memcpy(hp + o_plan_jsyn, job_syns.data(), job_syns.size() * 4);
	if (!subs.empty()) std::memcpy(hp + o_plan_subs, subs.data(), subs.size() * sizeof(rxgpu::FtPosSubterm));
	std::memcpy(hp + o_plan_terms, tcfg.data(), tcfg.size() * sizeof(rxgpu::FtTermCfg));
	if (!merge_grid.empty()) std::memcpy(hp + o_plan_mgrid, merge_grid.data(), merge_grid.size() * sizeof(rxgpu::FtGridEntry));

	if (h->clean_dirty) {
		RX_HIP(hipMemsetAsync(cbase, 0, h->d_clean.bytes, st));
		h->clean_dirty = false;
	}
	void* hp_dev = nullptr;   // the pinned staging buffer as the device sees it
	RX_HIP(hipHostGetDevicePointer(&hp_dev, hp, 0));

	rxgpu::FtPlan& p = job.p;
	p = rxgpu::FtPlan{};
	p.subs = reinterpret_cast<const rxgpu::FtPosSubterm*>(base + o_plan_subs);
	p.terms = reinterpret_cast<const rxgpu::FtTermCfg*>(base + o_plan_terms);
	p.merge_grid = reinterpret_cast<const rxgpu::FtGridEntry*>(base + o_plan_mgrid);
	p.n_merge_entries = uint32_t(merge_grid.size());
	p.merge_blocks = uint32_t(merge_blocks);
	p.nterms = nplan_terms;
	p.n_parts = nparts;
	p.n_part_qp = n_part_qp;
	p.syns = reinterpret_cast<const rxgpu::FtSynonym*>(base + o_plan_syns);
	p.n_syn = nsyn;
	p.syn_jobs = reinterpret_cast<const rxgpu::FtSynMaskJob*>(base + o_plan_jobs);
	p.job_syns = reinterpret_cast<const uint32_t*>(base + o_plan_jsyn);
	p.n_syn_jobs = uint32_t(syn_jobs.size());
	p.query_len = query_len;
	p.n_rows = n_rows;
	p.n_subs = uint32_t(subs.size());
	p.total_docs = N;
	p.nwords = nwords;
	p.max_merged = uint32_t(max_merged);
	p.merge_limit = cfg->merge_limit;
	p.simple = simple ? 1 : 0;
	p.prescore = prescore ? 1 : 0;
	p.check_removed = 1;
	p.distance_weight = float(cfg->distance_weight);
	p.distance_boost = float(cfg->distance_boost);
	p.full_match_boost = cfg->full_match_boost;
	p.removed = h->d_removed;
	p.excluded = d_excluded;
	p.mask = reinterpret_cast<uint32_t*>(base + o_mask);
	p.score = prescore ? reinterpret_cast<uint16_t*>(base + o_score) : nullptr;
	p.hist = prescore ? reinterpret_cast<uint32_t*>(cbase + o_hist) : nullptr;
	p.lookback_pre = prescore ? reinterpret_cast<unsigned long long*>(cbase + o_lb_pre) : nullptr;
	p.b_rec = reinterpret_cast<uint4*>(base + o_brec);
	p.bucket_off = reinterpret_cast<uint32_t*>(base + o_boff);
	p.bucket_cnt = reinterpret_cast<uint32_t*>(cbase + o_bcnt);
	p.adders = reinterpret_cast<uint32_t*>(base + o_adders);
	p.n_ranges = n_ranges;
	p.e_rank = reinterpret_cast<float*>(cbase + o_erank);
	p.e_idx = reinterpret_cast<uint32_t*>(base + o_eidx);
	p.e_field = reinterpret_cast<uint8_t*>(base + o_efield);
	p.sync = reinterpret_cast<uint32_t*>(cbase + o_sync);
	static const char* const stamps_env = std::getenv("RXGPU_FT_STAMPS");   // (a debugging hook: read once, not once per merge)
	p.dbg = stamps_env ? reinterpret_cast<unsigned long long*>(cbase + o_dbg) : nullptr;
	p.dbg_block = stamps_env ? uint32_t(std::atoi(stamps_env)) : 0;
	p.out_header = reinterpret_cast<uint32_t*>(ob);
	p.host_out = hp_dev;
	p.out_doc = reinterpret_cast<uint32_t*>(ob + align256(16));
	p.out_proc = reinterpret_cast<float*>(ob + align256(16) + align256(M * 4));
	p.out_terms_counter = reinterpret_cast<uint16_t*>(ob + align256(16) + 2 * align256(M * 4));
	p.out_field = reinterpret_cast<uint8_t*>(ob + align256(16) + 2 * align256(M * 4) + align256(M * 2));
	p.sparse = sparse ? 1 : 0;
	p.removed_bits = h->d_removed_bits;
	p.excluded_bits = d_excluded_bits;
	p.lb_units = reinterpret_cast<unsigned long long*>(cbase + o_lb_units);
	if (sparse) {
		p.t_doc = reinterpret_cast<uint32_t*>(base + o_tdoc);
		p.t_pos = reinterpret_cast<uint32_t*>(base + o_tpos);
		p.t_idx = reinterpret_cast<uint32_t*>(base + o_tidx);
		for (size_t si = 0; si < subs.size(); ++si) {   // what the unit kernels read of a sub-term, and its attribute word (ft_sparse.hip)
			const rxgpu::FtPosSubterm& ft = subs[si];
			const rxgpu::FtTermCfg& tc = tcfg[ft.term];
			const QueryTermIn& qt = terms[parts[ft.term].t_begin];
			// calcTermScores (mergerimpl.h:312-315): every field has the same boost, so maxBoostFromFields is field 0's
			const float proc = ft.proc * qt.opts->field_boost[0] * tc.opts_boost;
			uint32_t p16 = uint32_t(int32_t(proc)) & 0xFFFFu;   // static_cast<uint16_t>(float) as x86 evaluates it
			p16 = std::min<uint32_t>(p16, 65535u / 4);
			uint32_t attr = p16 | (ft.row << 20);
			if (si == tc.sub_begin) attr |= 1u << 16;
			if (tc.op == 2) attr |= 1u << 17;
			if (tc.op == 3) attr |= 1u << 18;
			p.sp_sub[si].doc = ft.doc;
			p.sp_sub[si].range_off = ft.range_off;
			p.sp_sub[si].n = uint32_t(ft.n);
			p.sp_sub[si].n_ranges = ft.n_ranges;
			p.sp_sub[si].attr = attr;
		}
		for (uint32_t pi = 0; pi < nparts && !simple; ++pi) {   // an AND term without postings empties the mask (buildRestrictingBitmask)
			if (tcfg[pi].op == 2 && tcfg[pi].sub_begin == tcfg[pi].sub_end) p.sp_empty_and = 1;
		}
	}
	if (h->sh_total > 1) {   // a document-range shard: its own ranges, the facts that span the shards arrive between the kernels
		// (multi-word synonyms are fine: their masks, term counts and the "only parts of a synonym" marks are facts of ONE document, and a
		// document lies in one shard — ft_syn_masks sees this shard's fragments, the caller drops the marked documents after the union)
		RX_CHECK(!resident, RXGPU_ERR_LOGIC, std::string(who) + ": a sharded ft index merges into the caller's lists (no resident results)");
		RX_CHECK(n_phrases == 0 || phrase_mode == 2, RXGPU_ERR_LOGIC, std::string(who) + ": a shard's phrases are run by the sharded layer");
		p.range_begin = h->sh_range_begin;
		p.range_count = h->sh_range_count;
		p.shard_index = h->sh_index;
		p.n_shards = h->sh_total;
		p.shard_hist = prescore ? h->sh_hist : nullptr;
		p.shard_pos = h->sh_pos;
		RX_HIP(hipMemsetAsync(p.adders, 0, std::max<size_t>(1, size_t(n_rows) * n_ranges) * 4, st));   // the other shards' columns
		RX_HIP(hipMemsetAsync(p.out_doc, 0xFF, M * 4, st));                                              // slots another shard fills stay marked
	}
	if (max_areas) {
		job.area_hdr_bytes = align256(M * nf * 2 * sizeof(uint32_t));
		job.area_bytes = M * nf * size_t(max_areas) * 3 * sizeof(uint32_t);
		if (int rc = h->d_areas.ensure(job.area_hdr_bytes + job.area_bytes); rc) return rc;
		RX_HIP(hipMemsetAsync(h->d_areas.ptr, 0, job.area_hdr_bytes, st));
		p.max_areas = max_areas;
		p.area_fields = nf;
		p.area_hdr = static_cast<uint32_t*>(h->d_areas.ptr);
		p.out_areas = reinterpret_cast<uint32_t*>(static_cast<char*>(h->d_areas.ptr) + job.area_hdr_bytes);
	}

	std::memcpy(hp + o_plan_self, &p, sizeof(p));
	job.d_plan = reinterpret_cast<const rxgpu::FtPlan*>(base + o_plan_self);
	job.dev_base = base;
	job.max_merged = max_merged;
	job.merged_postings = merged_postings;
	job.plan_bytes = plan_bytes;
	job.hp_dev = hp_dev;
	job.nsyn = nsyn;
	if (import_now) RX_HIP(rxgpu::launch_ft_import(hp_dev, base, plan_bytes, st));   // plan_bytes is a multiple of 256
	h->trace_us[1] += since(t_stage);
	return RXGPU_OK;
}

// Second half: the merged documents out of the lane's pinned staging buffer (ft_export wrote them there; the stream has been waited for).
int collect_merge(rxgpu_ft_index* h, const MergeJob& job, uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter, uint64_t* out_n,
				  int32_t* out_preselected, const char* who) {
	const char* hp = static_cast<const char*>(h->h_pinned);
	const size_t M = size_t(job.max_merged);
	const uint32_t* hdr = reinterpret_cast<const uint32_t*>(hp);
	RX_CHECK(hdr[1] == 0, RXGPU_ERR_DEVICE, std::string(who) + ": ordered look-back timed out on the device");
	const uint64_t n = hdr[0];
	RX_CHECK(n <= job.max_merged, RXGPU_ERR_DEVICE, std::string(who) + ": corrupt result header");
	h->clean_dirty = false;   // the merge ran to its end: ft_adders / ft_finish handed the tables back zeroed
	uint64_t kept = n;
	if (n && job.nsyn) {   // the documents that hold only parts of a multi-word synonym go (mergerimpl.h:533-555): the rest keeps its order
		const uint32_t* sd = reinterpret_cast<const uint32_t*>(hp + align256(16));
		const float* sp = reinterpret_cast<const float*>(hp + align256(16) + align256(M * 4));
		const uint16_t* st_ = reinterpret_cast<const uint16_t*>(hp + align256(16) + 2 * align256(M * 4));
		const uint8_t* sf = reinterpret_cast<const uint8_t*>(hp + align256(16) + 2 * align256(M * 4) + align256(M * 2));
		kept = 0;
		for (uint64_t i = 0; i < n; ++i) {
			if (st_[i] == 0xFFFFu) continue;
			out_doc[kept] = sd[i];
			out_proc[kept] = sp[i];
			if (out_terms_counter) out_terms_counter[kept] = st_[i];
			out_field[kept] = sf[i];
			++kept;
		}
	} else if (n) {
		std::memcpy(out_doc, hp + align256(16), n * 4);
		std::memcpy(out_proc, hp + align256(16) + align256(M * 4), n * 4);
		if (out_terms_counter) std::memcpy(out_terms_counter, hp + align256(16) + 2 * align256(M * 4), n * 2);
		std::memcpy(out_field, hp + align256(16) + 2 * align256(M * 4) + align256(M * 2), n);
	}
	*out_n = kept;
	if (out_preselected) *out_preselected = hdr[2] ? 1 : 0;
	return RXGPU_OK;
}

void ft_shards_destroy(rxgpu_ft_shard_set* ss) {
	if (!ss) return;
	for (rxgpu_ft_index* sh : ss->shards) rxgpu_ft_destroy(sh);
	for (uint32_t r = 0; r < ss->nranks; ++r) {
		(void)hipSetDevice(ss->rank_dev[r]);
		if (r < ss->rstream.size() && ss->rstream[r]) (void)hipStreamDestroy(ss->rstream[r]);
		if (r < ss->ev_rank.size() && ss->ev_rank[r]) (void)hipEventDestroy(ss->ev_rank[r]);
		if (r < ss->d_pos.size() && ss->d_pos[r]) (void)hipFree(ss->d_pos[r]);
		for (int k = 0; k < 2; ++k) {
			if (r < ss->d_send[k].size()) ss->d_send[k][r].release();
			if (r < ss->d_recv[k].size()) ss->d_recv[k][r].release();
		}
	}
	for (size_t s = 0; s < ss->ev_shard.size(); ++s) {
		if (ss->ev_shard[s]) {
			(void)hipSetDevice(ss->devices[s]);
			(void)hipEventDestroy(ss->ev_shard[s]);
		}
	}
	delete ss;
}

// Piece `k` (0 histograms, 1 tables) of every shard -> every rank's receive buffer.  Every shard's producer has been enqueued on its own
// stream and wrote bytes (a multiple of 4) at send_ptr(k, s); consumers enqueued afterwards on the shards' streams read recv_ptr(k, s).
int ft_shards_gather(rxgpu_ft_shard_set* ss, int k, size_t bytes) {
	const size_t S = ss->shards.size();
	for (size_t s = 0; s < S; ++s) {
		RX_HIP(hipSetDevice(ss->devices[s]));
		RX_HIP(hipEventRecord(ss->ev_shard[s], ss->shards[s]->stream));
	}
	for (uint32_t r = 0; r < ss->nranks; ++r) {
		RX_HIP(hipSetDevice(ss->rank_dev[r]));
		for (size_t s = 0; s < S; ++s) {
			if (ss->shard_rank[s] == r) RX_HIP(hipStreamWaitEvent(ss->rstream[r], ss->ev_shard[s], 0));
		}
	}
	if (ss->cs) {
		const rxgpu::RcclApi& api = rxgpu::rccl_api();
		std::lock_guard<std::mutex> lk(ss->cs->mtx);
		ncclResult_t nr = api.ncclGroupStart();
		for (uint32_t r = 0; r < ss->nranks && nr == ncclSuccess; ++r) {
			nr = api.ncclAllGather(ss->d_send[k][r].ptr, ss->d_recv[k][r].ptr, bytes / 4 * ss->slots, ncclUint32, ss->cs->comms[r], ss->rstream[r]);
		}
		const ncclResult_t ne = api.ncclGroupEnd();
		if (nr == ncclSuccess) nr = ne;
		if (nr != ncclSuccess) {
			set_error(std::string("sharded ft index: ncclAllGather: ") + api.ncclGetErrorString(nr));
			return RXGPU_ERR_DEVICE;
		}
		++ss->collectives;
	} else if (!ss->host_exchange && ss->nranks == 1) {
		// every shard lives on ONE device: the all-gather of a single rank is a copy on that device's exchange stream (no communicator is made
		// for one rank; with several devices the branch above runs — the same call pattern as the float_vector shards' exchange)
		RX_HIP(hipSetDevice(ss->rank_dev[0]));
		RX_HIP(hipMemcpyAsync(ss->d_recv[k][0].ptr, ss->d_send[k][0].ptr, bytes * ss->slots, hipMemcpyDeviceToDevice, ss->rstream[0]));
		++ss->collectives;
	} else {   // asked for, or no RCCL on this node: the same pieces through the host
		std::vector<char> all(size_t(ss->nranks) * ss->slots * bytes);
		for (uint32_t r = 0; r < ss->nranks; ++r) {
			RX_HIP(hipSetDevice(ss->rank_dev[r]));
			RX_HIP(hipMemcpyAsync(all.data() + size_t(r) * ss->slots * bytes, ss->d_send[k][r].ptr, ss->slots * bytes, hipMemcpyDeviceToHost, ss->rstream[r]));
		}
		for (uint32_t r = 0; r < ss->nranks; ++r) {
			RX_HIP(hipSetDevice(ss->rank_dev[r]));
			RX_HIP(hipStreamSynchronize(ss->rstream[r]));
		}
		for (uint32_t r = 0; r < ss->nranks; ++r) {
			RX_HIP(hipSetDevice(ss->rank_dev[r]));
			RX_HIP(hipMemcpyAsync(ss->d_recv[k][r].ptr, all.data(), all.size(), hipMemcpyHostToDevice, ss->rstream[r]));
			RX_HIP(hipStreamSynchronize(ss->rstream[r]));   // (`all` goes out of scope)
		}
	}
	for (uint32_t r = 0; r < ss->nranks; ++r) {
		RX_HIP(hipSetDevice(ss->rank_dev[r]));
		RX_HIP(hipEventRecord(ss->ev_rank[r], ss->rstream[r]));
	}
	for (size_t s = 0; s < S; ++s) {
		RX_HIP(hipSetDevice(ss->devices[s]));
		RX_HIP(hipStreamWaitEvent(ss->shards[s]->stream, ss->ev_rank[ss->shard_rank[s]], 0));
	}
	return RXGPU_OK;
}

int ft_shards_buffers(rxgpu_ft_shard_set* ss, int k, size_t bytes) {
	for (uint32_t r = 0; r < ss->nranks; ++r) {
		RX_HIP(hipSetDevice(ss->rank_dev[r]));
		const bool grow = ss->d_send[k][r].bytes < bytes * ss->slots;
		if (int rc = ss->d_send[k][r].ensure(bytes * ss->slots); rc) return rc;
		if (int rc = ss->d_recv[k][r].ensure(bytes * ss->slots * ss->nranks); rc) return rc;
		if (grow) RX_HIP(hipMemset(ss->d_send[k][r].ptr, 0, ss->d_send[k][r].bytes));   // padded slots (a device with fewer shards) stay zero
	}
	return RXGPU_OK;
}
inline char* ft_send_ptr(rxgpu_ft_shard_set* ss, int k, size_t s, size_t bytes) {
	return static_cast<char*>(ss->d_send[k][ss->shard_rank[s]].ptr) + size_t(ss->shard_slot[s]) * bytes;
}

// One merge over all shards (the caller holds the sharded handle's mutex): the ordinary launch train in its three pieces, the two exchanges
// between them, every shard's packed result, the slot-wise union.
int run_merge_sharded(rxgpu_ft_index* parent, const rxgpu_ft_config* cfg, bool simple, const std::vector<QueryTermIn>& terms, const uint32_t* word_ids,
					  const float* procs, const uint8_t* excluded, uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter,
					  uint64_t cap, uint64_t* out_n, int32_t* out_preselected, const char* who, const SynonymsIn* synonyms = nullptr,
					  const AreasOut* areas = nullptr) {
	rxgpu_ft_shard_set* ss = parent->shard_set;
	const uint32_t max_areas = areas ? areas->max_areas : 0u;
	const size_t S = ss->shards.size();
	RX_CHECK(ss->n_ranges > 0, RXGPU_ERR_LOGIC, std::string(who) + ": rxgpu_ft_set_docs was not called");
	int prev_dev = -1;
	(void)hipGetDevice(&prev_dev);
	struct Restore {
		int d;
		~Restore() { if (d >= 0) (void)hipSetDevice(d); }
	} restore{prev_dev};
	// exchange buffers first: the plans carry pointers into them
	const size_t fold_bytes = size_t(rxgpu::kFtFoldWords) * 4;
	const size_t nsubs = terms.empty() ? 0 : terms.back().sub_end;
	const size_t table_stride = std::max<size_t>(1, nsubs) * ss->n_ranges;   // >= rows x ranges of the plan (every sub-term is at most one row)
	if (int rc = ft_shards_buffers(ss, 0, fold_bytes); rc) return rc;
	if (int rc = ft_shards_buffers(ss, 1, table_stride * 4); rc) return rc;
	std::vector<MergeJob> jobs(S);
	std::vector<std::unique_lock<std::mutex>> locks;
	std::vector<std::shared_lock<std::shared_mutex>> dicts;
	std::vector<bool> active(S, false);
	bool empty = false;
	for (size_t s = 0; s < S; ++s) {
		locks.emplace_back(ss->shards[s]->mtx);
		dicts.emplace_back(ss->shards[s]->dict_mtx);
	}
	// Phrases first, on every shard (Merger::init, merger.h:73-81): a phrase is decided inside a document, so every shard runs PhraseMerger over
	// its own fragments; what spans the shards is NumDocsMerged() — the 2-phase estimate takes the sum — and the numbering of the phrase's rows.
	bool any_phrase = false;
	for (const QueryTermIn& t : terms) any_phrase = any_phrase || t.phrase_num >= 0;
	std::vector<std::vector<PhraseRows>> phrases(S);
	if (any_phrase) {
		for (size_t s = 0; s < S; ++s) {
			RX_HIP(hipSetDevice(ss->devices[s]));
			MergeJob scratch;
			if (int rc = prepare_merge(ss->shards[s], ss->shards[s]->stream, cfg, simple, terms, word_ids, procs, excluded, true, cfg->merge_limit, who, false, synonyms,
									   scratch, true, max_areas, &phrases[s], 1);
				rc)
				return rc;
			if (scratch.empty) return RXGPU_OK;   // min(mergeLimit, totalORVids) == 0: alike on every shard
		}
		// the admission cut of the whole index (phrasemerger.h:341): the first mergeLimit candidates in (row, document) order — row by row,
		// inside a row shard after shard (a shard's documents lie before the next one's).  What a shard keeps is a prefix of its own slots.
		for (size_t pi = 0; pi < phrases[0].size(); ++pi) {
			size_t n_rows = 0;
			bool any_pending = false;
			for (size_t s = 0; s < S; ++s) {
				RX_CHECK(phrases[s].size() == phrases[0].size(), RXGPU_ERR_LOGIC, std::string(who) + ": the shards disagree on the parts of the query");
				const PhraseRows& pr = phrases[s][pi];
				const size_t r = pr.pending ? pr.pending->row_admitted.size() : pr.rows.size();
				RX_CHECK(s == 0 || r == n_rows, RXGPU_ERR_LOGIC, std::string(who) + ": the shards disagree on the rows of a phrase");
				n_rows = r;
				any_pending = any_pending || pr.pending;
			}
			if (!any_pending) continue;
			std::vector<std::vector<uint32_t>> counts(S);
			for (size_t s = 0; s < S; ++s) {
				if (phrases[s][pi].pending) counts[s] = phrases[s][pi].pending->row_admitted;
			}
			const std::vector<uint64_t> keep = rxgpu::ft_shard_phrase_cut(counts, n_rows, cfg->merge_limit);
			for (size_t s = 0; s < S; ++s) {
				PhraseRows& pr = phrases[s][pi];
				if (!pr.pending) continue;
				pr.pending->admitted = uint32_t(std::min<uint64_t>(pr.pending->admitted, keep[s]));
				RX_HIP(hipSetDevice(ss->devices[s]));
				if (int rc = finish_phrase(ss->shards[s], procs, *pr.pending, pr, who); rc) return rc;
				pr.pending.reset();
			}
		}
		for (size_t pi = 0; pi < phrases[0].size(); ++pi) {
			uint64_t admitted = 0;
			for (size_t s = 0; s < S; ++s) {
				RX_CHECK(phrases[s].size() == phrases[0].size() && phrases[s][pi].rows.size() == phrases[0][pi].rows.size(), RXGPU_ERR_LOGIC,
						 std::string(who) + ": the shards disagree on the rows of a phrase");
				admitted += phrases[s][pi].admitted;
			}
			for (size_t s = 0; s < S; ++s) phrases[s][pi].admitted = uint32_t(std::min<uint64_t>(admitted, 0xFFFFFFFFull));
		}
	}
	for (size_t s = 0; s < S; ++s) {
		rxgpu_ft_index* sh = ss->shards[s];
		RX_HIP(hipSetDevice(ss->devices[s]));
		active[s] = sh->sh_range_count != 0;
		sh->sh_hist = static_cast<const uint32_t*>(ss->d_recv[0][ss->shard_rank[s]].ptr);
		sh->sh_pos = ss->d_pos[ss->shard_rank[s]];
		if (int rc = prepare_merge(sh, sh->stream, cfg, simple, terms, word_ids, procs, excluded, true, cfg->merge_limit, who, false, synonyms, jobs[s], true, max_areas,
								   any_phrase ? &phrases[s] : nullptr, any_phrase ? 2 : 0);
			rc)
			return rc;
		empty = empty || jobs[s].empty;
		sh->clean_dirty = !jobs[s].empty;   // an error return from here on leaves the kept-clean tables in an unknown state
	}
	if (empty) return RXGPU_OK;   // min(mergeLimit, totalORVids) == 0 — decided on the whole index's counts, alike on every shard
	const uint64_t M = jobs[0].max_merged;
	RX_CHECK(cap >= M, RXGPU_ERR_OVERFLOW, std::string(who) + ": output buffers too small");
	const bool prescore = jobs[0].p.prescore != 0;
	auto phase = [&](int ph) -> int {
		for (size_t s = 0; s < S; ++s) {
			if (!active[s]) continue;
			RX_HIP(hipSetDevice(ss->devices[s]));
			RX_HIP(rxgpu::launch_ft_merge_phase(jobs[s].d_plan, &jobs[s].p, 1, ph, ss->shards[s]->stream));
		}
		return RXGPU_OK;
	};
	if (int rc = phase(0); rc) return rc;
	if (prescore) {   // the histogram + popcount of every shard -> the sums; gate, threshold and tie quota are the whole index's
		for (size_t s = 0; s < S; ++s) {
			RX_HIP(hipSetDevice(ss->devices[s]));
			uint32_t* dst = reinterpret_cast<uint32_t*>(ft_send_ptr(ss, 0, s, fold_bytes));
			if (active[s]) {
				rxgpu::launch_ft_shard_fold(jobs[s].d_plan, dst, ss->shards[s]->stream);
			} else {
				RX_HIP(hipMemsetAsync(dst, 0, fold_bytes, ss->shards[s]->stream));
			}
		}
		if (int rc = ft_shards_gather(ss, 0, fold_bytes); rc) return rc;
		for (size_t s = 0; s < S; ++s) {
			if (!active[s]) continue;
			RX_HIP(hipSetDevice(ss->devices[s]));
			rxgpu::launch_ft_shard_hist_combine(jobs[s].d_plan, static_cast<const uint32_t*>(ss->d_recv[0][ss->shard_rank[s]].ptr), ss->d_pos[ss->shard_rank[s]],
												uint32_t(S), ss->shards[s]->stream);
		}
	}
	if (int rc = phase(1); rc) return rc;
	{   // the adder tables: every shard's own columns -> the table of the whole index
		const size_t n_table = size_t(jobs[0].p.n_rows) * jobs[0].p.n_ranges;
		for (size_t s = 0; s < S; ++s) {
			RX_HIP(hipSetDevice(ss->devices[s]));
			char* dst = ft_send_ptr(ss, 1, s, table_stride * 4);
			if (active[s] && n_table) {
				RX_HIP(hipMemcpyAsync(dst, jobs[s].p.adders, n_table * 4, hipMemcpyDeviceToDevice, ss->shards[s]->stream));
			} else {
				RX_HIP(hipMemsetAsync(dst, 0, std::max<size_t>(4, n_table * 4), ss->shards[s]->stream));
			}
		}
		if (int rc = ft_shards_gather(ss, 1, table_stride * 4); rc) return rc;
		for (size_t s = 0; s < S; ++s) {
			if (!active[s]) continue;
			RX_HIP(hipSetDevice(ss->devices[s]));
			rxgpu::launch_ft_shard_table_sum(jobs[s].p.adders, static_cast<const uint32_t*>(ss->d_recv[1][ss->shard_rank[s]].ptr), ss->d_pos[ss->shard_rank[s]], uint32_t(S),
											 n_table, table_stride, ss->shards[s]->stream);
		}
	}
	if (int rc = phase(2); rc) return rc;
	for (size_t s = 0; s < S; ++s) {
		if (!active[s]) continue;
		RX_HIP(hipSetDevice(ss->devices[s]));
		RX_HIP(rxgpu::launch_ft_export(jobs[s].d_plan, &jobs[s].p, 1, ss->shards[s]->stream));
	}
	// MergeDataAreas: {held, insertions} per (merge slot, field) and the areas as every shard's replay left them — a document's areas are
	// built where the document lies, at its GLOBAL merge slot, so the caller's arrays are the slot-wise union too
	const size_t nf = parent->num_fields;
	std::vector<std::vector<uint32_t>> area_hdr(areas ? S : 0), area_data(areas ? S : 0);
	for (size_t s = 0; s < S && areas; ++s) {
		if (!active[s]) continue;
		RX_HIP(hipSetDevice(ss->devices[s]));
		area_hdr[s].resize(size_t(M) * nf * 2);
		area_data[s].resize(jobs[s].area_bytes / sizeof(uint32_t));
		RX_HIP(hipMemcpyAsync(area_hdr[s].data(), jobs[s].p.area_hdr, area_hdr[s].size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ss->shards[s]->stream));
		RX_HIP(hipMemcpyAsync(area_data[s].data(), jobs[s].p.out_areas, jobs[s].area_bytes, hipMemcpyDeviceToHost, ss->shards[s]->stream));
	}
	for (size_t s = 0; s < S; ++s) {
		RX_HIP(hipSetDevice(ss->devices[s]));
		RX_HIP(hipStreamSynchronize(ss->shards[s]->stream));
	}
	++ss->merges;
	// ---- the slot-wise union: every merge slot was written by exactly one shard (the others left their 0xFFFFFFFF mark)
	uint64_t n = 0;
	bool have_n = false;
	int32_t presel = 0;
	for (size_t s = 0; s < S; ++s) {
		if (!active[s]) continue;
		const uint32_t* hdr = static_cast<const uint32_t*>(ss->shards[s]->h_pinned);
		RX_CHECK(hdr[1] == 0, RXGPU_ERR_DEVICE, std::string(who) + ": ordered look-back timed out on the device");
		RX_CHECK(!have_n || hdr[0] == n, RXGPU_ERR_DEVICE, std::string(who) + ": the shards disagree on the number of merged documents");
		n = hdr[0];
		have_n = true;
		presel = presel || hdr[2];
		ss->shards[s]->clean_dirty = false;
		ss->shards[s]->stat_postings += jobs[s].merged_postings;
	}
	RX_CHECK(n <= M, RXGPU_ERR_DEVICE, std::string(who) + ": corrupt result header");
	std::vector<uint8_t> filled(n, 0);
	for (size_t s = 0; s < S; ++s) {
		if (!active[s]) continue;
		const char* hp = static_cast<const char*>(ss->shards[s]->h_pinned);
		const uint32_t* sd = reinterpret_cast<const uint32_t*>(hp + align256(16));
		const float* sp = reinterpret_cast<const float*>(hp + align256(16) + align256(M * 4));
		const uint16_t* st_ = reinterpret_cast<const uint16_t*>(hp + align256(16) + 2 * align256(M * 4));
		const uint8_t* sf = reinterpret_cast<const uint8_t*>(hp + align256(16) + 2 * align256(M * 4) + align256(M * 2));
		for (uint64_t i = 0; i < n; ++i) {
			if (sd[i] == 0xFFFFFFFFu) continue;
			RX_CHECK(!filled[i], RXGPU_ERR_DEVICE, std::string(who) + ": two shards wrote one merge slot");
			filled[i] = 1;
			out_doc[i] = sd[i];
			out_proc[i] = sp[i];
			if (out_terms_counter) out_terms_counter[i] = st_[i];
			out_field[i] = sf[i];
			if (areas) {
				const size_t per_doc = nf * size_t(max_areas) * 3;
				for (size_t f = 0; f < nf; ++f) areas->cnt[i * nf + f] = area_hdr[s][(i * nf + f) * 2];
				std::memcpy(areas->areas + i * per_doc, area_data[s].data() + i * per_doc, per_doc * sizeof(uint32_t));
			}
		}
	}
	for (uint64_t i = 0; i < n; ++i) RX_CHECK(filled[i], RXGPU_ERR_DEVICE, std::string(who) + ": a merge slot no shard wrote");
	if (n && jobs[0].nsyn && out_terms_counter) {   // the documents that hold only parts of a multi-word synonym go (mergerimpl.h:533-555), as in collect_merge
		uint64_t kept = 0;
		for (uint64_t i = 0; i < n; ++i) {
			if (out_terms_counter[i] == 0xFFFFu) continue;
			out_doc[kept] = out_doc[i];
			out_proc[kept] = out_proc[i];
			out_terms_counter[kept] = out_terms_counter[i];
			out_field[kept] = out_field[i];
			++kept;
		}
		n = kept;
	}
	*out_n = n;
	if (out_preselected) *out_preselected = presel;
	return RXGPU_OK;
}


// Shared implementation of rxgpu_ft_merge_simple_raw / rxgpu_ft_merge_terms_raw.  out_terms_counter may be null (simple).
int run_merge(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, bool simple, const std::vector<QueryTermIn>& terms, const uint32_t* word_ids,
			  const float* procs, const uint8_t* excluded, uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter,
			  uint64_t cap, uint64_t* out_n, int32_t* out_preselected, const char* who, bool resident = false, const SynonymsIn* synonyms = nullptr,
			  const AreasOut* areas = nullptr) {
	using clk = std::chrono::steady_clock;
	if (h->shard_set) {   // document-range shards: the same train on every shard, two exchanges between its pieces
		RX_CHECK(!resident, RXGPU_ERR_LOGIC, std::string(who) + ": a sharded ft index merges into the caller's lists (no resident results)");
		RX_CHECK(out_doc && out_proc && out_field && (simple || out_terms_counter), RXGPU_ERR_OVERFLOW, std::string(who) + ": output buffers too small");
		return run_merge_sharded(h, cfg, simple, terms, word_ids, procs, excluded, out_doc, out_proc, out_field, out_terms_counter, cap, out_n, out_preselected, who,
								 synonyms, areas);
	}
	if (int rc = finish_pending(h, who); rc) return rc;
	auto since = [](clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); };
	hipStream_t st = h->stream;
	MergeJob job;
	if (int rc = prepare_merge(h, st, cfg, simple, terms, word_ids, procs, excluded, out_doc && out_proc && out_field && (simple || out_terms_counter), cap, who,
							   resident, synonyms, job, true, areas ? areas->max_areas : 0u);
		rc)
		return rc;
	if (job.empty) return RXGPU_OK;
	const rxgpu::FtPlan& p = job.p;
	const uint64_t max_merged = job.max_merged, merged_postings = job.merged_postings;
	const auto t_launch = clk::now();
	if (!h->ev_a) {
		RX_HIP(hipEventCreate(&h->ev_a));
		RX_HIP(hipEventCreate(&h->ev_b));
	}
	// from here on an error return leaves the kept-clean tables in an unknown state: the next merge clears them first
	h->clean_dirty = true;
	RX_HIP(hipEventRecord(h->ev_a, st));
	if (job.p.sparse) {
		RX_HIP(rxgpu::launch_ft_merge_sparse(job.d_plan, &job.p, 1, st));
	} else {
		RX_HIP(rxgpu::launch_ft_merge(job.d_plan, &job.p, 1, st));
	}
	RX_HIP(hipEventRecord(h->ev_b, st));
	(h->root ? h->root : h)->trains_dense += job.p.sparse ? 0 : 1;
	(h->root ? h->root : h)->trains_sparse += job.p.sparse ? 1 : 0;
	if (resident) {   // the result stays where ft_finish wrote it (d_out): the fusion kernel reads it there, nothing travels
		h->res_pending = true;
		h->res_has_syn = job.nsyn != 0;   // the fusion skips the documents ft_finish marked (they hold only parts of a synonym)
		h->prep_done = false;
		h->res_cap = uint32_t(max_merged);
		h->stat_postings += merged_postings;
		h->trace_us[2] += since(t_launch);
		h->trace_us[5] += 1;
		return RXGPU_OK;
	}
	RX_HIP(rxgpu::launch_ft_export(job.d_plan, &job.p, 1, st));
	std::vector<uint32_t> area_hdr;
	if (areas) {   // {held, insertions} per (document, field) and the areas, as the replay left them (the wait below covers the copies)
		area_hdr.resize(size_t(max_merged) * h->num_fields * 2);
		RX_HIP(hipMemcpyAsync(area_hdr.data(), job.p.area_hdr, area_hdr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		RX_HIP(hipMemcpyAsync(areas->areas, job.p.out_areas, job.area_bytes, hipMemcpyDeviceToHost, st));
	}
	h->trace_us[2] += since(t_launch);
	const auto t_wait = clk::now();
	// (the result is already on its way: ft_export, the last kernel of the train, writes it into the pinned staging buffer)
	{
		// a merge is ~0.1 ms of device time: poll for its end instead of sleeping in hipStreamSynchronize (the wake-up alone is tens of
		// microseconds); anything that takes longer than a few milliseconds falls back to the blocking wait
		const auto t_poll = clk::now();
		hipError_t q = hipStreamQuery(st);
		while (q == hipErrorNotReady && since(t_poll) < 3000.0) q = hipStreamQuery(st);
		if (q == hipErrorNotReady) {
			RX_HIP(hipStreamSynchronize(st));
		} else {
			RX_HIP(q);
		}
	}
	if (p.dbg) {
		unsigned long long raw[64];
		RX_HIP(hipMemcpy(raw, p.dbg, sizeof(raw), hipMemcpyDeviceToHost));
		RX_HIP(hipMemset(p.dbg, 0, sizeof(raw)));
		const int groups[][2] = {{0, 16}, {16, 24}, {24, 32}, {32, 48}};
		for (const auto& g : groups) {
			for (int k = g[0]; k < g[1]; ++k) {
				if (raw[k] && raw[g[0]]) h->stamps[k] += double(raw[k] - raw[g[0]]) * 0.01;   // 100 MHz -> us
			}
		}
	}
	h->trace_us[3] += since(t_wait);
	const auto t_unpack = clk::now();
	float ms = 0.f;
	(void)hipEventElapsedTime(&ms, h->ev_a, h->ev_b);
	h->stat_postings += merged_postings;
	h->stat_ms += ms;
	if (int rc = collect_merge(h, job, out_doc, out_proc, out_field, out_terms_counter, out_n, out_preselected, who); rc) return rc;
	if (areas) {
		RX_HIP(hipStreamSynchronize(st));   // (the polling above may have ended on the export kernel: the two copies behind it too, now)
		const size_t nf = h->num_fields;
		for (uint64_t i = 0; i < *out_n; ++i) {
			for (size_t f = 0; f < nf; ++f) areas->cnt[i * nf + f] = area_hdr[(i * nf + f) * 2];
		}
	}
	h->trace_us[4] += since(t_unpack);
	h->trace_us[5] += 1;
	return RXGPU_OK;
}

}  // namespace

int rxgpu_ft_merge_simple_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_term_opts* opts, uint32_t nsub,
							  const uint32_t* word_ids, const float* procs, const uint8_t* excluded, uint32_t* out_doc, float* out_proc,
							  uint8_t* out_field, uint64_t cap, uint64_t* out_n) {
	RX_CHECK(h && cfg && opts && out_n, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: null argument");
	*out_n = 0;
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, "rxgpu_ft_merge_simple_raw: rxgpu_ft_set_docs was not called");
	if (nsub == 0) return RXGPU_OK;
	RX_CHECK(word_ids && procs && opts->field_boost && opts->need_sum_rank, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_raw: null argument");
	LaneLock ll;
	if (int rc = checkout_lane(h, ll); rc) return rc;
	DevGuard dg(h->device);
	std::vector<QueryTermIn> terms{QueryTermIn{1, opts, 0, nsub}};
	return run_merge(ll.lane, cfg, true, terms, word_ids, procs, excluded, out_doc, out_proc, out_field, nullptr, cap, out_n, nullptr,
					 "rxgpu_ft_merge_simple_raw");
}

int rxgpu_ft_set_word_positions(rxgpu_ft_index* h, uint32_t word_id, uint64_t n, const uint32_t* doc, const uint32_t* pos_off, const uint64_t* fpos) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null ft index");
	RX_CHECK(n == 0 || (doc && pos_off && fpos), RXGPU_ERR_PARAMS, "rxgpu_ft_set_word_positions: null argument");
	if (h->shard_set) return ft_shards_set_word(h, word_id, n, doc, nullptr, nullptr, nullptr, nullptr, pos_off, fpos);
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
	std::unique_lock<std::shared_mutex> dict_lk(h->dict_mtx);
	DevGuard dg(h->device);
	rxgpu_ft_word& w = h->words[word_id];
	if (int rc = upload(w.pos_off, pos_off, n + 1); rc) return rc;
	if (int rc = upload(w.fpos, fpos, size_t(pos_off[n])); rc) return rc;
	return RXGPU_OK;
}

int rxgpu_ft_set_words_packed(rxgpu_ft_index* h, uint32_t nwords, const uint32_t* word_ids, const uint64_t* byte_off, const uint8_t* bytes,
							  const uint64_t* array_found_pos) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null ft index");
	if (nwords == 0) return RXGPU_OK;
	RX_CHECK(word_ids && byte_off && array_found_pos, RXGPU_ERR_PARAMS, "rxgpu_ft_set_words_packed: null argument");
	RX_CHECK(byte_off[0] == 0, RXGPU_ERR_PARAMS, "rxgpu_ft_set_words_packed: byte_off[0] must be 0");
	for (uint32_t w = 0; w < nwords; ++w) RX_CHECK(byte_off[w + 1] >= byte_off[w], RXGPU_ERR_PARAMS, "rxgpu_ft_set_words_packed: byte_off must not descend");
	RX_CHECK(byte_off[nwords] == 0 || bytes, RXGPU_ERR_PARAMS, "rxgpu_ft_set_words_packed: null argument");
	std::vector<const uint8_t*> data(nwords);
	std::vector<uint64_t> len(nwords);
	for (uint32_t w = 0; w < nwords; ++w) {
		data[w] = bytes + byte_off[w];
		len[w] = byte_off[w + 1] - byte_off[w];
	}
	return rxgpu_ft_set_words_packed_ptrs(h, nwords, word_ids, data.data(), len.data(), array_found_pos);
}

// The same with every word's stream where the caller keeps it (PackedIdRelVec::RawData() of each dictionary entry — separate allocations):
// the streams are gathered ONCE, in launch order, straight into the pinned staging buffer, and travel in one asynchronous copy.
int rxgpu_ft_set_words_packed_ptrs(rxgpu_ft_index* h, uint32_t nwords, const uint32_t* word_ids, const uint8_t* const* data, const uint64_t* len,
								   const uint64_t* array_found_pos) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null ft index");
	if (nwords == 0) return RXGPU_OK;
	RX_CHECK(word_ids && data && len && array_found_pos, RXGPU_ERR_PARAMS, "rxgpu_ft_set_words_packed_ptrs: null argument");
	const auto t_call = std::chrono::steady_clock::now();
	struct WallClock {
		rxgpu_ft_index* h;
		std::chrono::steady_clock::time_point t0;
		~WallClock() { h->packed_wall_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
	};
	uint64_t total_bytes = 0;
	for (uint32_t w = 0; w < nwords; ++w) {
		RX_CHECK(len[w] == 0 || data[w], RXGPU_ERR_PARAMS, "rxgpu_ft_set_words_packed_ptrs: null stream");
		total_bytes += len[w];
	}
	std::lock_guard<std::mutex> lk(h->mtx);
	WallClock wall{h, t_call};
	std::unique_lock<std::shared_mutex> dict_lk(h->dict_mtx);   // no merge on any lane reads the dictionary meanwhile
	DevGuard dg(h->device);
	RX_HIP(hipStreamSynchronize(h->stream));
	// wavefronts of similar work: the words are launched longest first (a wavefront lasts as long as its longest stream).  A bucket sort by
	// the length's power of two is enough for that — O(n); a comparison sort of a 100 000-word dictionary cost 8 ms of this call.
	std::vector<uint32_t> order(nwords);
	{
		uint32_t bucket_n[65] = {0};
		auto bucket_of = [&](uint32_t w) {
			const uint64_t l = len[w];
			return l ? 64 - uint32_t(__builtin_clzll(l)) : 0u;   // 0 .. 64
		};
		for (uint32_t w = 0; w < nwords; ++w) bucket_n[bucket_of(w)] += 1;
		uint32_t start[65];
		uint32_t at = 0;
		for (int b = 64; b >= 0; --b) {
			start[b] = at;
			at += bucket_n[b];
		}
		for (uint32_t w = 0; w < nwords; ++w) order[start[bucket_of(w)]++] = w;
	}
	// (start, end) of every stream in the staging buffer, launch order; scratch of the call (streams, offsets, counts, pieces, slices) lives
	// in buffers the index keeps and grows: no hipMalloc / hipFree pair — each a device synchronisation — per call
	std::vector<uint64_t> off(size_t(nwords) * 2), afp(nwords);
	{
		uint64_t at = 0;
		for (uint32_t k = 0; k < nwords; ++k) {
			const uint32_t w = order[k];
			off[2 * size_t(k)] = at;
			at += len[w];
			off[2 * size_t(k) + 1] = at;
			afp[k] = array_found_pos[w];
		}
	}
	auto len_of = [&](uint32_t k) { return off[2 * size_t(k) + 1] - off[2 * size_t(k)]; };
	// one wavefront per word (ft_packed_wave); RXGPU_FT_PACKED_THREAD=1: the one-thread-per-word kernels of round 2 (cross-check, comparison)
	const bool wave = std::getenv("RXGPU_FT_PACKED_THREAD") == nullptr;
	// pieces of kFtPackedSegBytes: the counting pass leaves a checkpoint in each, the writing pass runs one wavefront per piece
	uint32_t nsegs = 0;
	std::vector<uint32_t> seg_first;
	if (wave) {
		seg_first.resize(size_t(nwords) + 1);
		seg_first[0] = 0;
		for (uint32_t k = 0; k < nwords; ++k) {
			const uint64_t len = len_of(k);
			const uint64_t pieces = std::max<uint64_t>(1, (len + rxgpu::kFtPackedSegBytes - 1) / rxgpu::kFtPackedSegBytes);
			RX_CHECK(seg_first[k] + pieces < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "rxgpu_ft_set_words_packed: too many stream bytes in one call");
			seg_first[k + 1] = uint32_t(seg_first[k] + pieces);
		}
		nsegs = seg_first[nwords];
	}
	// pinned staging: [streams | (start, end) pairs | array_found_pos | piece -> word | first piece of a word]; everything but the streams
	// travels first (one copy), the streams follow in chunks so that the gather of the next chunk, the copy of this one and the counting
	// pass of the previous one overlap
	const size_t o_off = align256(size_t(total_bytes) + 16), o_afp = o_off + align256(size_t(nwords) * 16), o_sw = o_afp + align256(size_t(nwords) * 8);
	const size_t o_sf = o_sw + align256(size_t(nsegs) * 4), in_bytes = o_sf + align256((size_t(nwords) + 1) * 4);
	if (int rc = h->d_pk_in.ensure(in_bytes); rc) return rc;
	if (int rc = h->ensure_pinned(in_bytes); rc) return rc;
	uint8_t* hp = static_cast<uint8_t*>(h->h_pinned);
	std::memcpy(hp + o_off, off.data(), size_t(nwords) * 16);
	std::memcpy(hp + o_afp, afp.data(), size_t(nwords) * 8);
	if (wave) {
		uint32_t* sw = reinterpret_cast<uint32_t*>(hp + o_sw);
		for (uint32_t k = 0; k < nwords; ++k) std::fill(sw + seg_first[k], sw + seg_first[k + 1], k);
		std::memcpy(hp + o_sf, seg_first.data(), (size_t(nwords) + 1) * 4);
	}
	uint8_t* d_bytes = static_cast<uint8_t*>(h->d_pk_in.ptr);
	uint64_t* d_off = reinterpret_cast<uint64_t*>(d_bytes + o_off);
	uint64_t* d_afp = reinterpret_cast<uint64_t*>(d_bytes + o_afp);
	RX_HIP(hipMemcpyAsync(d_bytes + o_off, hp + o_off, in_bytes - o_off, hipMemcpyHostToDevice, h->stream));
	if (int rc = h->d_pk_cnt.ensure(size_t(nwords) * sizeof(rxgpu::FtPackedCounts)); rc) return rc;
	rxgpu::FtPackedCounts* d_counts = static_cast<rxgpu::FtPackedCounts*>(h->d_pk_cnt.ptr);
	rxgpu::FtPackedSegs segs{};
	if (wave) {
		if (int rc = h->d_pk_segs.ensure(size_t(nsegs) * sizeof(rxgpu::FtPackedCheckpoint)); rc) return rc;
		RX_HIP(hipMemsetAsync(h->d_pk_segs.ptr, 0xFF, size_t(nsegs) * sizeof(rxgpu::FtPackedCheckpoint), h->stream));
		segs.seg_word = reinterpret_cast<const uint32_t*>(d_bytes + o_sw);
		segs.seg_first = reinterpret_cast<const uint32_t*>(d_bytes + o_sf);
		segs.cps = static_cast<rxgpu::FtPackedCheckpoint*>(h->d_pk_segs.ptr);
		segs.nsegs = nsegs;
	}
	// The streams may travel in chunks of whole words (launch order: the longest words first) so that the gather of chunk c + 1, the copy of
	// chunk c and the counting pass over chunk c - 1 overlap.  Measured on the 100 000-word dictionary of tools/bench_ft_packed.py: 49.8 ms
	// per call with 8 MB chunks against 34.6 ms in one piece (profiles/rd4k_ft_packed_chunked.json, rd4h_ft_packed.json) — a chunk's
	// counting pass lasts as long as its longest stream, and the longest streams are what travels first.  One piece is the default;
	// RXGPU_FT_PACKED_CHUNK_MB=<n> cuts.
	std::vector<uint32_t> chunk_first{0u};
	{
		uint64_t target = ~0ull;
		if (const char* e = std::getenv("RXGPU_FT_PACKED_CHUNK_MB")) {
			if (std::atol(e) > 0) target = uint64_t(std::atol(e)) << 20;
		}
		uint64_t acc = 0;
		for (uint32_t k = 0; k < nwords; ++k) {
			acc += len_of(k);
			if (acc >= target && k + 1 < nwords) {
				chunk_first.push_back(k + 1);
				acc = 0;
			}
		}
		chunk_first.push_back(nwords);
	}
	const uint32_t nchunks = uint32_t(chunk_first.size() - 1);
	// the gather: one pass over the streams by a few threads (100 000 pieces of a few hundred bytes: one thread moves ~7 GB/s of them), chunk
	// by chunk; the calling thread sends a chunk on its way as soon as every worker is through with it
	const unsigned nthr = total_bytes > (8u << 20) ? 4u : 1u;
	std::vector<std::atomic<uint32_t>> chunk_done(nchunks);
	for (auto& c : chunk_done) c.store(0, std::memory_order_relaxed);
	auto gather = [&](unsigned t) {
		for (uint32_t c = 0; c < nchunks; ++c) {
			const uint32_t k0 = chunk_first[c], kn = chunk_first[c + 1] - k0;
			for (uint32_t k = k0 + uint32_t(uint64_t(kn) * t / nthr), e = k0 + uint32_t(uint64_t(kn) * (t + 1) / nthr); k < e; ++k) {
				const uint64_t n = len_of(k);
				if (n) std::memcpy(hp + off[2 * size_t(k)], data[order[k]], size_t(n));
			}
			chunk_done[c].fetch_add(1, std::memory_order_release);
		}
	};
	std::memset(hp + total_bytes, 0, 16);
	std::vector<std::thread> gatherers;
	struct Joiner {
		std::vector<std::thread>& threads;
		~Joiner() {
			for (std::thread& t : threads) t.join();
		}
	} joiner{gatherers};
	if (nthr > 1) {
		for (unsigned t = 0; t < nthr; ++t) gatherers.emplace_back(gather, t);
	}
	// The counting pass of a chunk needs only that chunk's bytes, and lasts as long as the chunk's longest stream (a serial walk): on ONE
	// stream the chunks' kernels ran one behind the other and the pass took 34 ms instead of 6.  They run on four streams, each behind
	// its chunk's copy, and overlap like the wavefronts of a single launch do.
	for (hipStream_t& ps : h->pk_streams) {
		if (!ps) RX_HIP(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
	}
	EventPair ev_count, ev_write;
	if (int rc = ev_count.create(); rc) return rc;
	if (int rc = ev_write.create(); rc) return rc;
	struct EventList {
		std::vector<hipEvent_t> v;
		~EventList() {
			for (hipEvent_t e : v) (void)hipEventDestroy(e);
		}
		int add(hipEvent_t* out) {
			hipEvent_t e = nullptr;
			RX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			v.push_back(e);
			*out = e;
			return RXGPU_OK;
		}
	} chunk_events;
	std::vector<hipEvent_t> counted(nchunks, nullptr);
	RX_HIP(hipEventRecord(ev_count.a, h->stream));   // (behind the setup copies: the pass is timed from here to its last kernel, uploads included)
	for (uint32_t c = 0; c < nchunks; ++c) {
		if (nthr > 1) {
			while (chunk_done[c].load(std::memory_order_acquire) < nthr) std::this_thread::yield();
		} else if (c == 0) {
			gather(0);   // small calls: everything at once on this thread
		}
		const uint32_t k0 = chunk_first[c], k1 = chunk_first[c + 1];
		const uint64_t b0 = off[2 * size_t(k0)], b1 = off[2 * size_t(k1 - 1) + 1] + (c + 1 == nchunks ? 16 : 0);
		if (b1 > b0) RX_HIP(hipMemcpyAsync(d_bytes + b0, hp + b0, size_t(b1 - b0), hipMemcpyHostToDevice, h->stream));
		hipEvent_t landed = nullptr;
		if (int rc = chunk_events.add(&landed); rc) return rc;
		RX_HIP(hipEventRecord(landed, h->stream));
		hipStream_t ks = h->pk_streams[c % 4];
		RX_HIP(hipStreamWaitEvent(ks, landed, 0));
		if (wave) {
			RX_HIP(rxgpu::launch_ft_packed_count(d_bytes, d_off, d_afp, nwords, h->num_fields, d_counts, &segs, ks, k0, k1 - k0));
		} else if (c + 1 == nchunks) {   // the thread-per-word kernels: one launch over all words
			RX_HIP(rxgpu::launch_ft_packed_count(d_bytes, d_off, d_afp, nwords, h->num_fields, d_counts, nullptr, ks, 0, nwords));
		}
		if (int rc = chunk_events.add(&counted[c]); rc) return rc;
		RX_HIP(hipEventRecord(counted[c], ks));
	}
	for (uint32_t c = 0; c < nchunks; ++c) RX_HIP(hipStreamWaitEvent(h->stream, counted[c], 0));
	RX_HIP(hipEventRecord(ev_count.b, h->stream));
	std::vector<rxgpu::FtPackedCounts> counts(nwords);
	RX_HIP(hipMemcpyAsync(counts.data(), d_counts, size_t(nwords) * sizeof(rxgpu::FtPackedCounts), hipMemcpyDeviceToHost, h->stream));
	RX_HIP(hipStreamSynchronize(h->stream));
	auto status_text = [](uint32_t st) {
		switch (st) {
			case rxgpu::kFtPackedTruncated: return "truncated varint stream";
			case rxgpu::kFtPackedDocOrder: return "document ids must ascend strictly";
			case rxgpu::kFtPackedField: return "field out of range";
			default: return "posting list too long";
		}
	};
	for (uint32_t k = 0; k < nwords; ++k) {
		RX_CHECK(counts[k].status == rxgpu::kFtPackedOk, RXGPU_ERR_PARAMS,
				 std::string("rxgpu_ft_set_words_packed: word ") + std::to_string(word_ids[order[k]]) + ": " + status_text(counts[k].status));
	}
	// one pool for the whole batch, every array of every word on a 256-byte boundary (the kernels read document ids 16 bytes at a time)
	Carver cv;
	std::vector<rxgpu::FtPackedOut> outs(nwords);
	struct Slices {
		size_t doc, pos_off, fpos, ent_off, ent_field, ent_tf, ent_first, range_off;
	};
	std::vector<Slices> sl(nwords);
	for (uint32_t k = 0; k < nwords; ++k) {
		const rxgpu::FtPackedCounts& c = counts[k];
		if (!c.n) continue;
		sl[k].doc = cv.take(size_t(c.n) * 4);
		sl[k].pos_off = cv.take((size_t(c.n) + 1) * 4);
		sl[k].fpos = cv.take(size_t(c.npos) * 8);
		sl[k].ent_off = cv.take((size_t(c.n) + 1) * 4);
		sl[k].ent_field = cv.take(size_t(c.nent));
		sl[k].ent_tf = cv.take(size_t(c.nent) * 4);
		sl[k].ent_first = cv.take(size_t(c.nent) * 4);
		outs[k].n_ranges = c.last_doc / rxgpu::kFtRangeDocs + 2;
		sl[k].range_off = cv.take(size_t(outs[k].n_ranges) * 4);
	}
	std::shared_ptr<void> pool;
	char* base = nullptr;
	if (cv.off) {
		void* raw = nullptr;
		RX_HIP(hipMalloc(&raw, cv.off));
		const int device = h->device;
		pool = std::shared_ptr<void>(raw, [device](void* q) {
			DevGuard g(device);
			(void)hipFree(q);
		});
		base = static_cast<char*>(raw);
	}
	for (uint32_t k = 0; k < nwords; ++k) {
		if (!counts[k].n) continue;
		outs[k].doc = reinterpret_cast<uint32_t*>(base + sl[k].doc);
		outs[k].pos_off = reinterpret_cast<uint32_t*>(base + sl[k].pos_off);
		outs[k].fpos = reinterpret_cast<uint64_t*>(base + sl[k].fpos);
		outs[k].ent_off = reinterpret_cast<uint32_t*>(base + sl[k].ent_off);
		outs[k].ent_field = reinterpret_cast<uint8_t*>(base + sl[k].ent_field);
		outs[k].ent_tf = reinterpret_cast<uint32_t*>(base + sl[k].ent_tf);
		outs[k].ent_first_pos = reinterpret_cast<uint32_t*>(base + sl[k].ent_first);
		outs[k].range_off = reinterpret_cast<uint32_t*>(base + sl[k].range_off);
	}
	if (int rc = h->d_pk_outs.ensure(size_t(nwords) * sizeof(rxgpu::FtPackedOut)); rc) return rc;
	RX_HIP(hipMemcpyAsync(h->d_pk_outs.ptr, outs.data(), size_t(nwords) * sizeof(rxgpu::FtPackedOut), hipMemcpyHostToDevice, h->stream));
	RX_HIP(hipEventRecord(ev_write.a, h->stream));
	RX_HIP(rxgpu::launch_ft_packed_write(d_bytes, d_off, d_afp, nwords, h->num_fields, static_cast<const rxgpu::FtPackedOut*>(h->d_pk_outs.ptr), d_counts, wave ? &segs : nullptr, h->stream));
	RX_HIP(hipEventRecord(ev_write.b, h->stream));
	std::vector<rxgpu::FtPackedCounts> again(nwords);
	RX_HIP(hipMemcpyAsync(again.data(), d_counts, size_t(nwords) * sizeof(rxgpu::FtPackedCounts), hipMemcpyDeviceToHost, h->stream));
	// the dictionary entries while the write pass runs (100 000 map insertions are a fifth of this call); should the pass disagree with the
	// counting pass below — an internal error — the words of the call are left empty
	h->words.reserve(h->words.size() + nwords);
	for (uint32_t k = 0; k < nwords; ++k) {
		rxgpu_ft_word& w = h->words[word_ids[order[k]]];
		w.release();
		const rxgpu::FtPackedCounts& c = counts[k];
		if (!c.n) continue;
		w.n = c.n;
		w.nent = c.nent;
		w.last_doc = c.last_doc;
		w.doc = outs[k].doc;
		w.ent_off = outs[k].ent_off;
		w.ent_field = outs[k].ent_field;
		w.ent_tf = outs[k].ent_tf;
		w.ent_first_pos = outs[k].ent_first_pos;
		w.pos_off = outs[k].pos_off;
		w.fpos = outs[k].fpos;
		w.range_off = outs[k].range_off;
		w.n_ranges = outs[k].n_ranges;
		w.pool = pool;
	}
	const hipError_t waited = hipStreamSynchronize(h->stream);
	bool agree = waited == hipSuccess;
	for (uint32_t k = 0; k < nwords && agree; ++k) {
		agree = again[k].status == rxgpu::kFtPackedOk && again[k].n == counts[k].n && again[k].npos == counts[k].npos && again[k].nent == counts[k].nent;
	}
	if (!agree) {
		for (uint32_t k = 0; k < nwords; ++k) h->words[word_ids[order[k]]].release();
		RX_HIP(waited);
		RX_CHECK(false, RXGPU_ERR_DEVICE, "rxgpu_ft_set_words_packed: the write pass disagrees with the counting pass");
	}
	h->packed_count_ms += ev_count.elapsed_ms();
	h->packed_write_ms += ev_write.elapsed_ms();
	h->packed_bytes_in += total_bytes;
	h->packed_bytes_out += cv.off;
	return RXGPU_OK;
}

int rxgpu_ft_read_packed_stats(rxgpu_ft_index* h, double* count_ms, double* write_ms, uint64_t* bytes_in, uint64_t* bytes_out) {
	RX_CHECK(h && count_ms && write_ms && bytes_in && bytes_out, RXGPU_ERR_PARAMS, "rxgpu_ft_read_packed_stats: null argument");
	std::lock_guard<std::mutex> lk(h->mtx);
	*count_ms = h->packed_count_ms;
	*write_ms = h->packed_write_ms;
	*bytes_in = h->packed_bytes_in;
	*bytes_out = h->packed_bytes_out;
	h->packed_count_ms = h->packed_write_ms = 0.0;
	h->packed_bytes_in = h->packed_bytes_out = 0;
	return RXGPU_OK;
}

int rxgpu_ft_read_packed_wall(rxgpu_ft_index* h, double* wall_ms) {
	RX_CHECK(h && wall_ms, RXGPU_ERR_PARAMS, "rxgpu_ft_read_packed_wall: null argument");
	std::lock_guard<std::mutex> lk(h->mtx);
	*wall_ms = h->packed_wall_ms;
	h->packed_wall_ms = 0.0;
	return RXGPU_OK;
}

int rxgpu_ft_word_df(rxgpu_ft_index* h, uint32_t word_id, uint64_t* out_df) {
	RX_CHECK(h && out_df, RXGPU_ERR_PARAMS, "rxgpu_ft_word_df: null argument");
	*out_df = 0;
	rxgpu_ft_index* src = h;
	if (h->shard_set) {   // every shard keeps the word's entry with the whole list's length
		if (h->shard_set->shards.empty()) return RXGPU_OK;
		src = h->shard_set->shards[0];
	}
	std::shared_lock<std::shared_mutex> dict_lk(src->dict_mtx);
	const auto it = src->words.find(word_id);
	if (it != src->words.end()) *out_df = word_df(it->second);
	return RXGPU_OK;
}

int rxgpu_ft_get_word(rxgpu_ft_index* h, uint32_t word_id, uint64_t* n, uint64_t* npos, uint64_t* nent, uint32_t* doc, uint32_t* pos_off, uint64_t* fpos,
					  uint32_t* ent_off, uint8_t* ent_field, uint32_t* ent_tf, uint32_t* ent_first_pos, uint32_t* n_ranges, uint32_t* range_off) {
	RX_CHECK(h && n && npos && nent && n_ranges, RXGPU_ERR_PARAMS, "rxgpu_ft_get_word: null argument");
	std::lock_guard<std::mutex> lk(h->mtx);
	std::shared_lock<std::shared_mutex> dict_lk(h->dict_mtx);
	DevGuard dg(h->device);
	RX_HIP(hipStreamSynchronize(h->stream));
	const auto it = h->words.find(word_id);
	RX_CHECK(it != h->words.end(), RXGPU_ERR_PARAMS, "rxgpu_ft_get_word: unknown word");
	const rxgpu_ft_word& w = it->second;
	*n = w.n;
	*nent = w.nent;
	*n_ranges = w.n_ranges;
	*npos = 0;
	if (!w.n) return RXGPU_OK;
	if (w.pos_off) {
		uint32_t last = 0;
		RX_HIP(hipMemcpy(&last, w.pos_off + w.n, 4, hipMemcpyDeviceToHost));
		*npos = last;
	}
	auto down = [](void* dst, const void* src, size_t bytes) -> int {
		if (dst && src && bytes) RX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
		return RXGPU_OK;
	};
	if (int rc = down(doc, w.doc, w.n * 4); rc) return rc;
	if (int rc = down(pos_off, w.pos_off, (w.n + 1) * 4); rc) return rc;
	if (int rc = down(fpos, w.fpos, size_t(*npos) * 8); rc) return rc;
	if (int rc = down(ent_off, w.ent_off, (w.n + 1) * 4); rc) return rc;
	if (int rc = down(ent_field, w.ent_field, w.nent); rc) return rc;
	if (int rc = down(ent_tf, w.ent_tf, w.nent * 4); rc) return rc;
	if (int rc = down(ent_first_pos, w.ent_first_pos, w.nent * 4); rc) return rc;
	if (int rc = down(range_off, w.range_off, size_t(w.n_ranges) * 4); rc) return rc;
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
	RX_CHECK(sub_off[nterms] == 0 || (word_ids && procs), RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_raw: null argument");
	LaneLock ll;
	if (int rc = checkout_lane(h, ll); rc) return rc;
	DevGuard dg(h->device);
	std::vector<QueryTermIn> terms(nterms);
	for (uint32_t t = 0; t < nterms; ++t) terms[t] = QueryTermIn{ops[t], &opts[t], sub_off[t], sub_off[t + 1]};
	return run_merge(ll.lane, cfg, false, terms, word_ids, procs, excluded, out_doc, out_proc, out_field, out_terms_counter, cap, out_n, out_preselected,
					 "rxgpu_ft_merge_terms_raw");
}

namespace {
// the checks shared by rxgpu_ft_merge_query_raw / _resident; *empty: QueryMergeData::Empty() (querymergedata.h:208)
int query_terms(const char* who, uint32_t nterms, const int32_t* ops, const rxgpu_ft_term_opts* opts, const int32_t* phrase_num, const int32_t* distance,
				const uint32_t* sub_off, const uint32_t* word_ids, const float* procs, std::vector<QueryTermIn>& terms, bool* empty, bool* simple) {
	*empty = true;
	*simple = false;
	if (nterms == 0) return RXGPU_OK;
	RX_CHECK(ops && opts && sub_off, RXGPU_ERR_PARAMS, std::string(who) + ": null argument");
	RX_CHECK(nterms < 0x7FFF, RXGPU_ERR_PARAMS, std::string(who) + ": too many terms");
	RX_CHECK(sub_off[nterms] == 0 || (word_ids && procs), RXGPU_ERR_PARAMS, std::string(who) + ": null argument");
	terms.resize(nterms);
	uint32_t nparts = 0;
	for (uint32_t t = 0; t < nterms; ++t) {
		RX_CHECK(ops[t] >= 1 && ops[t] <= 3, RXGPU_ERR_PARAMS, std::string(who) + ": op must be 1 (OR), 2 (AND) or 3 (NOT)");
		terms[t] = QueryTermIn{ops[t], &opts[t], sub_off[t], sub_off[t + 1], phrase_num ? phrase_num[t] : -1, distance ? distance[t] : 1};
		if (terms[t].phrase_num < 0 || t == 0 || terms[t - 1].phrase_num != terms[t].phrase_num) ++nparts;
	}
	*empty = nparts == 1 && ops[0] == 3;
	*simple = nparts == 1 && ops[0] != 3 && terms[0].phrase_num < 0;
	return RXGPU_OK;
}
}  // namespace

int rxgpu_ft_merge_query_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nterms, const int32_t* ops, const rxgpu_ft_term_opts* opts,
							 const int32_t* phrase_num, const int32_t* distance, const uint32_t* sub_off, const uint32_t* word_ids, const float* procs,
							 const uint8_t* excluded, uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter, uint64_t cap,
							 uint64_t* out_n, int32_t* out_preselected) {
	const char* who = "rxgpu_ft_merge_query_raw";
	RX_CHECK(h && cfg && out_n, RXGPU_ERR_PARAMS, std::string(who) + ": null argument");
	*out_n = 0;
	if (out_preselected) *out_preselected = 0;
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, std::string(who) + ": field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, std::string(who) + ": rxgpu_ft_set_docs was not called");
	std::vector<QueryTermIn> terms;
	bool empty = false, simple = false;
	if (int rc = query_terms(who, nterms, ops, opts, phrase_num, distance, sub_off, word_ids, procs, terms, &empty, &simple); rc) return rc;
	if (empty) return RXGPU_OK;
	LaneLock ll;
	if (int rc = checkout_lane(h, ll); rc) return rc;
	DevGuard dg(h->device);
	return run_merge(ll.lane, cfg, simple, terms, word_ids, procs, excluded, out_doc, out_proc, out_field, out_terms_counter, cap, out_n, out_preselected, who);
}

int rxgpu_ft_merge_query2_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_query* q, const uint8_t* excluded, uint32_t* out_doc, float* out_proc,
							  uint8_t* out_field, uint16_t* out_terms_counter, uint64_t cap, uint64_t* out_n, int32_t* out_preselected) {
	const char* who = "rxgpu_ft_merge_query2_raw";
	RX_CHECK(h && cfg && q && out_n, RXGPU_ERR_PARAMS, std::string(who) + ": null argument");
	*out_n = 0;
	if (out_preselected) *out_preselected = 0;
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, std::string(who) + ": field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, std::string(who) + ": rxgpu_ft_set_docs was not called");
	RX_CHECK(q->nsyn > 0 || q->nsyn_terms == 0, RXGPU_ERR_PARAMS, std::string(who) + ": synonym terms without synonyms");
	std::vector<QueryTermIn> terms;
	bool empty = false, simple = false;
	if (int rc = query_terms(who, q->nterms, q->ops, q->opts, q->phrase_num, q->distance, q->sub_off, q->word_ids, q->procs, terms, &empty, &simple); rc) return rc;
	if (empty) return RXGPU_OK;   // QueryMergeData::Empty() looks at the query parts only
	SynonymsIn syn;
	if (q->nsyn) {
		RX_CHECK(q->syn_term_off && q->part_syn_off, RXGPU_ERR_PARAMS, std::string(who) + ": null synonym tables");
		for (uint32_t k = 0; k < q->nsyn_terms; ++k) {
			const uint32_t t = q->nterms + k;
			RX_CHECK(q->ops[t] >= 1 && q->ops[t] <= 3, RXGPU_ERR_PARAMS, std::string(who) + ": op must be 1 (OR), 2 (AND) or 3 (NOT)");
			terms.push_back(QueryTermIn{q->ops[t], &q->opts[t], q->sub_off[t], q->sub_off[t + 1], -1, 1});
		}
		syn.nsyn = q->nsyn;
		syn.first_term = q->nterms;
		syn.syn_term_off = q->syn_term_off;
		syn.part_syn_off = q->part_syn_off;
		syn.part_syn = q->part_syn;
		syn.suppressed = q->suppressed;
		simple = false;
	}
	LaneLock ll;
	if (int rc = checkout_lane(h, ll); rc) return rc;
	DevGuard dg(h->device);
	return run_merge(ll.lane, cfg, simple, terms, q->word_ids, q->procs, excluded, out_doc, out_proc, out_field, out_terms_counter, cap, out_n, out_preselected, who,
					 false, q->nsyn ? &syn : nullptr);
}

// Merger<IdCont, MergeDataAreas<Area>, ...>::Merge (merger.h:36-57 with kWithRegularAreas): the merge of rxgpu_ft_merge_query2_raw plus, per merged
// document and field, the areas its postings left — what highlight() / snippet() read.
int rxgpu_ft_merge_query_areas_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_query* q, const uint8_t* excluded, uint32_t max_areas_in_doc,
								   uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter, uint64_t cap, uint64_t* out_n,
								   int32_t* out_preselected, uint32_t* out_area_cnt, uint32_t* out_areas) {
	const char* who = "rxgpu_ft_merge_query_areas_raw";
	RX_CHECK(h && cfg && q && out_n && out_area_cnt && out_areas, RXGPU_ERR_PARAMS, std::string(who) + ": null argument");
	*out_n = 0;
	if (out_preselected) *out_preselected = 0;
	RX_CHECK(max_areas_in_doc >= 1 && max_areas_in_doc <= 4096, RXGPU_ERR_PARAMS, std::string(who) + ": max_areas_in_doc must be in [1, 4096] (FTConfig::maxAreasInDoc; unlimited areas stay on the CPU merger)");
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, std::string(who) + ": field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, std::string(who) + ": rxgpu_ft_set_docs was not called");
	RX_CHECK(q->nsyn == 0 && q->nsyn_terms == 0, RXGPU_ERR_LOGIC, std::string(who) + ": areas are built for queries without multi-word synonyms");
	std::vector<QueryTermIn> terms;
	bool empty = false, simple = false;
	if (int rc = query_terms(who, q->nterms, q->ops, q->opts, q->phrase_num, q->distance, q->sub_off, q->word_ids, q->procs, terms, &empty, &simple); rc) return rc;
	if (empty) return RXGPU_OK;
	LaneLock ll;
	if (int rc = checkout_lane(h, ll); rc) return rc;
	DevGuard dg(h->device);
	AreasOut ao;
	ao.max_areas = max_areas_in_doc;
	ao.cnt = out_area_cnt;
	ao.areas = out_areas;
	return run_merge(ll.lane, cfg, simple, terms, q->word_ids, q->procs, excluded, out_doc, out_proc, out_field, out_terms_counter, cap, out_n, out_preselected, who,
					 false, nullptr, &ao);
}

// Q queries over one index in ONE launch train (ft_merge.hip: grid.y = query).  The launch floors and the ramp of every kernel's grid are
// paid once per train instead of once per merge, and the device sees Q x the work at a time: what a planner with several FT queries in
// hand (or the hybrid path with its batch of queries) calls instead of Q single merges.
int rxgpu_ft_merge_batch_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nq, const rxgpu_ft_query* queries, const uint8_t* const* excluded,
							 uint32_t* const* out_doc, float* const* out_proc, uint8_t* const* out_field, uint16_t* const* out_terms_counter, uint64_t cap,
							 uint64_t* out_n, int32_t* out_preselected) {
	const char* who = "rxgpu_ft_merge_batch_raw";
	RX_CHECK(h && cfg && out_n && (nq == 0 || (queries && out_doc && out_proc && out_field && out_terms_counter)), RXGPU_ERR_PARAMS, std::string(who) + ": null argument");
	for (uint32_t i = 0; i < nq; ++i) {
		out_n[i] = 0;
		if (out_preselected) out_preselected[i] = 0;
	}
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, std::string(who) + ": field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, std::string(who) + ": rxgpu_ft_set_docs was not called");
	// queries with phrases or multi-word synonyms have kernels of their own in front of the train (ft_phrase.hip, ft_syn_masks): one by one
	std::vector<uint32_t> batched;
	for (uint32_t i = 0; i < nq; ++i) {
		const rxgpu_ft_query& q = queries[i];
		bool plain = q.nsyn == 0 && q.nsyn_terms == 0;
		for (uint32_t t = 0; plain && q.phrase_num && t < q.nterms; ++t) plain = q.phrase_num[t] < 0;
		if (plain && !h->shard_set) {   // (a sharded index: every shard's handle runs one train and its exchanges at a time — the merges one by one)
			batched.push_back(i);
			continue;
		}
		if (int rc = rxgpu_ft_merge_query2_raw(h, cfg, &q, excluded ? excluded[i] : nullptr, out_doc[i], out_proc[i], out_field[i], out_terms_counter[i], cap, &out_n[i],
											   out_preselected ? &out_preselected[i] : nullptr);
			rc)
			return rc;
	}
	if (batched.empty()) return RXGPU_OK;
	std::lock_guard<std::mutex> batch_lk(h->batch_mtx);
	std::shared_lock<std::shared_mutex> dict_lk(h->dict_mtx);
	DevGuard dg(h->device);
	if (!h->batch_stream) RX_HIP(hipStreamCreateWithFlags(&h->batch_stream, hipStreamNonBlocking));
	if (!h->ev_ba) {
		RX_HIP(hipEventCreate(&h->ev_ba));
		RX_HIP(hipEventCreate(&h->ev_bb));
	}
	constexpr size_t kPlansBytes = (size_t(rxgpu::kFtBatchMax) * sizeof(rxgpu::FtPlan) + 255) & ~size_t(255);
	if (!h->h_batch_plans) RX_HIP(hipHostMalloc(&h->h_batch_plans, kPlansBytes, hipHostMallocDefault));
	if (int rc = h->d_batch_plans.ensure(kPlansBytes); rc) return rc;
	void* plans_dev_view = nullptr;
	RX_HIP(hipHostGetDevicePointer(&plans_dev_view, h->h_batch_plans, 0));
	hipStream_t st = h->batch_stream;
	using clk = std::chrono::steady_clock;
	for (size_t c0 = 0; c0 < batched.size(); c0 += rxgpu::kFtBatchMax) {
		const size_t c1 = std::min(batched.size(), c0 + rxgpu::kFtBatchMax);
		std::vector<MergeJob> jobs;
		std::vector<rxgpu_ft_index*> job_lane;
		std::vector<uint32_t> job_query;
		std::vector<rxgpu::FtPlan> host_plans;
		rxgpu::FtImportBatch pieces{};
		uint64_t postings = 0;
		for (size_t c = c0; c < c1; ++c) {
			const uint32_t i = batched[c];
			const rxgpu_ft_query& q = queries[i];
			std::vector<QueryTermIn> terms;
			bool empty = false, simple = false;
			if (int rc = query_terms(who, q.nterms, q.ops, q.opts, q.phrase_num, q.distance, q.sub_off, q.word_ids, q.procs, terms, &empty, &simple); rc) return rc;
			if (empty) continue;
			const size_t k = jobs.size();
			while (h->batch_lanes.size() <= k) {
				auto lane = std::make_unique<rxgpu_ft_index>();
				lane->device = h->device;
				lane->num_fields = h->num_fields;
				lane->root = h;
				h->batch_lanes.push_back(std::move(lane));
			}
			rxgpu_ft_index* lane = h->batch_lanes[k].get();
			lane->total_docs = h->total_docs;
			lane->d_words = h->d_words;
			lane->d_avg = h->d_avg;
			lane->d_removed = h->d_removed;
			lane->d_removed_bits = h->d_removed_bits;
			lane->h_avg = h->h_avg;
			MergeJob job;
			if (int rc = prepare_merge(lane, st, cfg, simple, terms, q.word_ids, q.procs, excluded ? excluded[i] : nullptr,
									   out_doc[i] && out_proc[i] && out_field[i] && (simple || out_terms_counter[i]), cap, who, false, nullptr, job, false);
				rc)
				return rc;
			if (job.empty) continue;
			pieces.src[k] = job.hp_dev;
			pieces.dst[k] = job.dev_base;
			pieces.n16[k] = uint32_t(job.plan_bytes / 16);
			postings += job.merged_postings;
			host_plans.push_back(job.p);
			jobs.push_back(job);
			job_lane.push_back(lane);
			job_query.push_back(i);
		}
		const uint32_t B = uint32_t(jobs.size());
		if (!B) continue;
		// the sparse train's queries in front, the dense train's behind: each train is launched over its own run of plans
		std::stable_sort(host_plans.begin(), host_plans.end(), [](const rxgpu::FtPlan& a, const rxgpu::FtPlan& b) {
			const int ka = a.sparse ? (a.prescore ? 0 : 1) : 2, kb = b.sparse ? (b.prescore ? 0 : 1) : 2;   // (the sparse train runs the preselecting queries as one run)
			return ka < kb;
		});
		uint32_t n_sparse = 0;
		while (n_sparse < B && host_plans[n_sparse].sparse) ++n_sparse;
		std::memcpy(h->h_batch_plans, host_plans.data(), size_t(B) * sizeof(rxgpu::FtPlan));
		pieces.src[B] = plans_dev_view;
		pieces.dst[B] = h->d_batch_plans.ptr;
		pieces.n16[B] = uint32_t((size_t(B) * sizeof(rxgpu::FtPlan) + 15) / 16);
		pieces.n = B + 1;
		const rxgpu::FtPlan* d_plans = static_cast<const rxgpu::FtPlan*>(h->d_batch_plans.ptr);
		for (rxgpu_ft_index* lane : job_lane) lane->clean_dirty = true;   // until the train has run to its end
		RX_HIP(rxgpu::launch_ft_import_batch(pieces, st));
		RX_HIP(hipEventRecord(h->ev_ba, st));
		RX_HIP(rxgpu::launch_ft_merge_sparse(d_plans, host_plans.data(), n_sparse, st));
		RX_HIP(rxgpu::launch_ft_merge(d_plans + n_sparse, host_plans.data() + n_sparse, B - n_sparse, st));
		RX_HIP(hipEventRecord(h->ev_bb, st));
		h->trains_sparse += n_sparse;
		h->trains_dense += B - n_sparse;
		RX_HIP(rxgpu::launch_ft_export(d_plans, host_plans.data(), B, st));
		{
			const auto t_poll = clk::now();
			hipError_t qs = hipStreamQuery(st);
			while (qs == hipErrorNotReady && std::chrono::duration<double, std::micro>(clk::now() - t_poll).count() < 3000.0) qs = hipStreamQuery(st);
			if (qs == hipErrorNotReady) {
				RX_HIP(hipStreamSynchronize(st));
			} else {
				RX_HIP(qs);
			}
		}
		float ms = 0.f;
		(void)hipEventElapsedTime(&ms, h->ev_ba, h->ev_bb);
		h->stat_postings += postings;
		h->stat_ms += ms;
		h->batch_trains += 1;
		h->batch_merges += B;
		for (uint32_t k = 0; k < B; ++k) {
			const uint32_t i = job_query[k];
			if (int rc = collect_merge(job_lane[k], jobs[k], out_doc[i], out_proc[i], out_field[i], out_terms_counter[i], &out_n[i],
									   out_preselected ? &out_preselected[i] : nullptr, who);
				rc)
				return rc;
		}
	}
	return RXGPU_OK;
}
void rxgpu_ft_set_train_mode(int mode) { ft_train_mode()->store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }
int rxgpu_ft_read_train_stats(rxgpu_ft_index* h, uint64_t* dense_merges, uint64_t* sparse_merges) {
	RX_CHECK(h && dense_merges && sparse_merges, RXGPU_ERR_PARAMS, "rxgpu_ft_read_train_stats: null argument");
	*dense_merges = h->trains_dense.exchange(0);
	*sparse_merges = h->trains_sparse.exchange(0);
	return RXGPU_OK;
}
int rxgpu_ft_read_batch_stats(rxgpu_ft_index* h, uint64_t* trains, uint64_t* merges) {
	RX_CHECK(h && trains && merges, RXGPU_ERR_PARAMS, "rxgpu_ft_read_batch_stats: null argument");
	std::lock_guard<std::mutex> lk(h->batch_mtx);
	*trains = h->batch_trains;
	*merges = h->batch_merges;
	return RXGPU_OK;
}

int rxgpu_ft_merge_query_resident(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nterms, const int32_t* ops, const rxgpu_ft_term_opts* opts,
								  const int32_t* phrase_num, const int32_t* distance, const uint32_t* sub_off, const uint32_t* word_ids, const float* procs,
								  const uint8_t* excluded, int32_t* out_enqueued) {
	const char* who = "rxgpu_ft_merge_query_resident";
	RX_CHECK(h && cfg && out_enqueued, RXGPU_ERR_PARAMS, std::string(who) + ": null argument");
	*out_enqueued = 0;
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, std::string(who) + ": field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, std::string(who) + ": rxgpu_ft_set_docs was not called");
	std::vector<QueryTermIn> terms;
	bool empty = false, simple = false;
	if (int rc = query_terms(who, nterms, ops, opts, phrase_num, distance, sub_off, word_ids, procs, terms, &empty, &simple); rc) return rc;
	std::unique_lock<std::mutex> lk(h->mtx);
	open_resident_session(h, lk);
	std::shared_lock<std::shared_mutex> dict_lk(h->dict_mtx);
	DevGuard dg(h->device);
	h->res_cap = 0;
	h->prep_done = false;
	if (empty) return finish_pending(h, who);   // nothing is merged: the fusion sees an empty FT side
	uint64_t n = 0;
	if (int rc = run_merge(h, cfg, simple, terms, word_ids, procs, excluded, nullptr, nullptr, nullptr, nullptr, 0, &n, nullptr, who, true); rc) return rc;
	*out_enqueued = h->res_pending ? 1 : 0;
	return RXGPU_OK;
}

// ... and the resident form of rxgpu_ft_merge_query2_raw: multi-word synonyms included.  The documents that hold only parts of a synonym
// stay in the result with their 0xFFFF mark; the fusion kernels treat them as absent (HybridFuseArgs::ft_terms).
int rxgpu_ft_merge_query2_resident(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_query* q, const uint8_t* excluded, int32_t* out_enqueued) {
	const char* who = "rxgpu_ft_merge_query2_resident";
	RX_CHECK(h && cfg && q && out_enqueued, RXGPU_ERR_PARAMS, std::string(who) + ": null argument");
	*out_enqueued = 0;
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, std::string(who) + ": field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, std::string(who) + ": rxgpu_ft_set_docs was not called");
	RX_CHECK(q->nsyn > 0 || q->nsyn_terms == 0, RXGPU_ERR_PARAMS, std::string(who) + ": synonym terms without synonyms");
	std::vector<QueryTermIn> terms;
	bool empty = false, simple = false;
	if (int rc = query_terms(who, q->nterms, q->ops, q->opts, q->phrase_num, q->distance, q->sub_off, q->word_ids, q->procs, terms, &empty, &simple); rc) return rc;
	SynonymsIn syn;
	if (!empty && q->nsyn) {
		RX_CHECK(q->syn_term_off && q->part_syn_off, RXGPU_ERR_PARAMS, std::string(who) + ": null synonym tables");
		for (uint32_t k = 0; k < q->nsyn_terms; ++k) {
			const uint32_t t = q->nterms + k;
			RX_CHECK(q->ops[t] >= 1 && q->ops[t] <= 3, RXGPU_ERR_PARAMS, std::string(who) + ": op must be 1 (OR), 2 (AND) or 3 (NOT)");
			terms.push_back(QueryTermIn{q->ops[t], &q->opts[t], q->sub_off[t], q->sub_off[t + 1], -1, 1});
		}
		syn.nsyn = q->nsyn;
		syn.first_term = q->nterms;
		syn.syn_term_off = q->syn_term_off;
		syn.part_syn_off = q->part_syn_off;
		syn.part_syn = q->part_syn;
		syn.suppressed = q->suppressed;
		simple = false;
	}
	std::unique_lock<std::mutex> lk(h->mtx);
	open_resident_session(h, lk);
	std::shared_lock<std::shared_mutex> dict_lk(h->dict_mtx);
	DevGuard dg(h->device);
	h->res_cap = 0;
	h->prep_done = false;
	if (empty) return finish_pending(h, who);   // nothing is merged: the fusion sees an empty FT side
	uint64_t n = 0;
	if (int rc = run_merge(h, cfg, simple, terms, q->word_ids, q->procs, excluded, nullptr, nullptr, nullptr, nullptr, 0, &n, nullptr, who, true, q->nsyn ? &syn : nullptr); rc) {
		return rc;
	}
	*out_enqueued = h->res_pending ? 1 : 0;
	return RXGPU_OK;
}

// ---------------------------------------------------------------------------------------------------- hybrid: merges that stay in HBM + the fusion
int rxgpu_ft_merge_simple_resident(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_term_opts* opts, uint32_t nsub,
								   const uint32_t* word_ids, const float* procs, const uint8_t* excluded) {
	RX_CHECK(h && cfg && opts, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_resident: null argument");
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_resident: field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, "rxgpu_ft_merge_simple_resident: rxgpu_ft_set_docs was not called");
	RX_CHECK(nsub > 0 && word_ids && procs && opts->field_boost && opts->need_sum_rank, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_simple_resident: null argument");
	std::unique_lock<std::mutex> lk(h->mtx);
	open_resident_session(h, lk);
	std::shared_lock<std::shared_mutex> dict_lk(h->dict_mtx);
	DevGuard dg(h->device);
	std::vector<QueryTermIn> terms{QueryTermIn{1, opts, 0, nsub}};
	uint64_t n = 0;
	h->res_cap = 0;
	h->prep_done = false;
	return run_merge(h, cfg, true, terms, word_ids, procs, excluded, nullptr, nullptr, nullptr, nullptr, 0, &n, nullptr, "rxgpu_ft_merge_simple_resident", true);
}

int rxgpu_ft_merge_terms_resident(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nterms, const int32_t* ops, const rxgpu_ft_term_opts* opts,
								  const uint32_t* sub_off, const uint32_t* word_ids, const float* procs, const uint8_t* excluded) {
	RX_CHECK(h && cfg && ops && opts && sub_off, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_resident: null argument");
	RX_CHECK(cfg->num_fields == h->num_fields, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_resident: field count mismatch");
	RX_CHECK(h->total_docs > 0, RXGPU_ERR_LOGIC, "rxgpu_ft_merge_terms_resident: rxgpu_ft_set_docs was not called");
	RX_CHECK(nterms >= 2 && nterms < 0xFFFF, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_resident: 2 or more terms (one term: rxgpu_ft_merge_simple_resident)");
	for (uint32_t t = 0; t < nterms; ++t) RX_CHECK(ops[t] >= 1 && ops[t] <= 3, RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_resident: op must be 1 (OR), 2 (AND) or 3 (NOT)");
	RX_CHECK(sub_off[nterms] == 0 || (word_ids && procs), RXGPU_ERR_PARAMS, "rxgpu_ft_merge_terms_resident: null argument");
	std::unique_lock<std::mutex> lk(h->mtx);
	open_resident_session(h, lk);
	std::shared_lock<std::shared_mutex> dict_lk(h->dict_mtx);
	DevGuard dg(h->device);
	std::vector<QueryTermIn> terms(nterms);
	for (uint32_t t = 0; t < nterms; ++t) terms[t] = QueryTermIn{ops[t], &opts[t], sub_off[t], sub_off[t + 1]};
	uint64_t n = 0;
	h->res_cap = 0;
	h->prep_done = false;
	return run_merge(h, cfg, false, terms, word_ids, procs, excluded, nullptr, nullptr, nullptr, nullptr, 0, &n, nullptr, "rxgpu_ft_merge_terms_resident", true);
}

namespace {
int check_hybrid_params(const rxgpu_hybrid_params* p, const char* who) {
	RX_CHECK(p, RXGPU_ERR_PARAMS, std::string(who) + ": null parameters");
	RX_CHECK(p->kind == 0 || p->kind == 1, RXGPU_ERR_PARAMS, std::string(who) + ": kind must be 0 (RRF) or 1 (linear)");
	return RXGPU_OK;
}
void fill_reranker(rxgpu::HybridFuseArgs& a, const rxgpu_hybrid_params* p, int metric) {
	a.kind = p->kind;
	a.is_union = p->is_union ? 1 : 0;
	a.desc = p->desc ? 1 : 0;
	for (int i = 0; i < 5; ++i) a.params[i] = p->params[i];
	a.metric_l2 = metric == RXGPU_METRIC_L2 ? 1 : 0;
}
}  // namespace

namespace {
// the FT-side arguments of the two fusion kernels for the resident merge of `h` (M = its max_merged; 0: no resident merge)
int fuse_ft_args(rxgpu_ft_index* h, uint32_t M, int32_t min_rank, const rxgpu_hybrid_params* params, int metric, const void* d_row_of_doc,
				 rxgpu::HybridFuseArgs& a) {
	const size_t key_bytes = align256(size_t(2) * std::max<uint32_t>(M, 1) * 4), cls_bytes = align256(size_t(2) * std::max<uint32_t>(M, 1) * 2);
	if (int rc = h->d_fuse.ensure(key_bytes + cls_bytes + align256(sizeof(rxgpu::HybridFuseState))); rc) return rc;
	char* ob = static_cast<char*>(h->d_out.ptr);
	if (M) {   // the packed layout run_merge gave d_out for max_merged = M
		a.ft_count_ptr = reinterpret_cast<const uint32_t*>(ob);
		a.ft_doc = reinterpret_cast<const uint32_t*>(ob + align256(16));
		a.ft_proc = reinterpret_cast<const float*>(ob + align256(16) + align256(size_t(M) * 4));
		if (h->res_has_syn) a.ft_terms = reinterpret_cast<const uint16_t*>(ob + align256(16) + 2 * align256(size_t(M) * 4));
	}
	a.ft_n = 0;
	a.ft_cap = M;
	a.min_rank = float(min_rank);
	a.row_of_doc = static_cast<const int32_t*>(d_row_of_doc);
	fill_reranker(a, params, metric);
	a.scratch_key = static_cast<uint32_t*>(h->d_fuse.ptr);
	a.scratch_cls = reinterpret_cast<uint16_t*>(static_cast<char*>(h->d_fuse.ptr) + key_bytes);
	a.state = reinterpret_cast<rxgpu::HybridFuseState*>(static_cast<char*>(h->d_fuse.ptr) + key_bytes + cls_bytes);
	return RXGPU_OK;
}
void prep_signature(int32_t min_rank, const rxgpu_hybrid_params* p, const void* d_row_of_doc, double sig[8]) {
	sig[0] = min_rank;
	sig[1] = p->kind * 4 + (p->desc ? 2 : 0);
	for (int i = 0; i < 5; ++i) sig[2 + i] = p->params[i];
	sig[7] = double(reinterpret_cast<uintptr_t>(d_row_of_doc));
}
int enqueue_prepare(rxgpu_ft_index* h, int32_t min_rank, const rxgpu_hybrid_params* params, int metric, const void* d_row_of_doc) {
	const uint32_t M = h->res_pending ? h->res_cap : 0;
	rxgpu::HybridFuseArgs a{};
	if (int rc = fuse_ft_args(h, M, min_rank, params, metric, d_row_of_doc, a); rc) return rc;
	if (!h->ev_pa) {
		RX_HIP(hipEventCreate(&h->ev_pa));
		RX_HIP(hipEventCreate(&h->ev_pb));
	}
	RX_HIP(hipEventRecord(h->ev_pa, h->stream));
	RX_HIP(rxgpu::launch_hybrid_prepare(a, h->stream));
	RX_HIP(hipEventRecord(h->ev_pb, h->stream));
	h->prep_timed = true;
	prep_signature(min_rank, params, d_row_of_doc, h->prep_sig);
	h->prep_done = true;
	return RXGPU_OK;
}
}  // namespace

// The FT-only half of the fusion (postProcessResults, id order, class / group tables), enqueued behind the resident merge so that it
// runs while the KNN search is still streaming the corpus.  Optional: rxgpu_hybrid_fuse_resident enqueues it itself when it was not.
int rxgpu_hybrid_prepare_resident(rxgpu_ft_index* h, int32_t min_rank, const rxgpu_hybrid_params* params, int metric, const void* d_row_of_doc) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "rxgpu_hybrid_prepare_resident: null argument");
	if (int rc = check_hybrid_params(params, "rxgpu_hybrid_prepare_resident"); rc) return rc;
	std::unique_lock<std::mutex> lk(h->mtx);
	if (int rc = check_resident_session(h, lk, "rxgpu_hybrid_prepare_resident"); rc) return rc;
	DevGuard dg(h->device);
	return enqueue_prepare(h, min_rank, params, metric, d_row_of_doc);
}

int rxgpu_hybrid_fuse_resident(rxgpu_ft_index* h, int32_t min_rank, const rxgpu_hybrid_params* params, int metric, const void* d_knn_dist,
							   const void* d_knn_row, const void* d_knn_count, uint32_t knn_n, uint32_t k, void* knn_stream, const void* d_row_of_doc,
							   const void* d_rowid_of_row, int32_t* out_ids, float* out_ranks, uint64_t cap, uint64_t* out_n, uint32_t* out_flags) {
	RX_CHECK(h && out_n, RXGPU_ERR_PARAMS, "rxgpu_hybrid_fuse_resident: null argument");
	*out_n = 0;
	if (out_flags) *out_flags = 0;
	if (int rc = check_hybrid_params(params, "rxgpu_hybrid_fuse_resident"); rc) return rc;
	RX_CHECK(k <= uint32_t(rxgpu::kMaxFuseKnn) && k <= knn_n, RXGPU_ERR_PARAMS, "rxgpu_hybrid_fuse_resident: k must be <= 1024 and <= the entries of the KNN list");
	RX_CHECK(knn_n == 0 || (d_knn_dist && d_knn_row), RXGPU_ERR_PARAMS, "rxgpu_hybrid_fuse_resident: null KNN list");
	std::unique_lock<std::mutex> lk(h->mtx);
	if (int rc = check_resident_session(h, lk, "rxgpu_hybrid_fuse_resident"); rc) return rc;
	struct SessionEnd {   // whatever happens below, this thread's session ends with its fusion
		rxgpu_ft_index* h;
		~SessionEnd() { close_resident_session(h); }
	} session_end{h};
	DevGuard dg(h->device);
	const uint32_t M = h->res_pending ? h->res_cap : 0;   // no resident merge: an empty FT side (the merge found nothing to do)
	const size_t out_cap = size_t(M) + k;
	RX_CHECK(cap >= out_cap && (out_cap == 0 || (out_ids && out_ranks)), RXGPU_ERR_OVERFLOW, "rxgpu_hybrid_fuse_resident: output buffers too small");
	{   // the FT-only half, unless the caller had it enqueued already (for exactly these parameters)
		double sig[8];
		prep_signature(min_rank, params, d_row_of_doc, sig);
		if (!h->prep_done || std::memcmp(sig, h->prep_sig, sizeof(sig)) != 0) {
			if (int rc = enqueue_prepare(h, min_rank, params, metric, d_row_of_doc); rc) return rc;
		}
	}
	// the result leaves through the pinned staging buffer: the kernel's stores go straight to host memory, no copy-engine start-up
	const size_t o_ids = 256, o_ranks = o_ids + align256(out_cap * 4), stage = o_ranks + align256(out_cap * 4);
	if (int rc = h->ensure_pinned(stage); rc) return rc;
	char* hp = static_cast<char*>(h->h_pinned);
	void* hp_dev = nullptr;
	RX_HIP(hipHostGetDevicePointer(&hp_dev, hp, 0));
	char* hd = static_cast<char*>(hp_dev);
	hipStream_t st = h->stream;
	if (knn_stream) {   // the KNN search ran on the caller's stream: the join waits for it on the device, the host does not
		if (!h->ev_knn) RX_HIP(hipEventCreateWithFlags(&h->ev_knn, hipEventDisableTiming));
		RX_HIP(hipEventRecord(h->ev_knn, static_cast<hipStream_t>(knn_stream)));
		RX_HIP(hipStreamWaitEvent(st, h->ev_knn, 0));
	}
	rxgpu::HybridFuseArgs a{};
	if (int rc = fuse_ft_args(h, M, min_rank, params, metric, d_row_of_doc, a); rc) return rc;
	a.knn_dist = static_cast<const float*>(d_knn_dist);
	a.knn_row = static_cast<const uint32_t*>(d_knn_row);
	a.knn_count_ptr = static_cast<const uint32_t*>(d_knn_count);
	a.knn_n = knn_n;
	a.k = k;
	a.knn_negate = metric == RXGPU_METRIC_L2 ? 0 : 1;
	a.rowid_of_row = static_cast<const int32_t*>(d_rowid_of_row);
	a.out_header = reinterpret_cast<uint32_t*>(hd);
	a.out_ids = reinterpret_cast<int32_t*>(hd + o_ids);
	a.out_ranks = reinterpret_cast<float*>(hd + o_ranks);
	static const bool stamps = std::getenv("RXGPU_FUSE_STAMPS") != nullptr;
	if (stamps) a.dbg = reinterpret_cast<unsigned long long*>(hd + 64);   // inside the 256-byte header region of the staging buffer
	if (!h->ev_fa) {
		RX_HIP(hipEventCreate(&h->ev_fa));
		RX_HIP(hipEventCreate(&h->ev_fb));
	}
	RX_HIP(hipEventRecord(h->ev_fa, st));
	RX_HIP(rxgpu::launch_hybrid_join(a, st));
	RX_HIP(hipEventRecord(h->ev_fb, st));
	h->prep_done = false;
	{
		using clk = std::chrono::steady_clock;
		const auto t_poll = clk::now();
		hipError_t q = hipStreamQuery(st);
		while (q == hipErrorNotReady && std::chrono::duration<double, std::micro>(clk::now() - t_poll).count() < 3000.0) q = hipStreamQuery(st);
		if (q == hipErrorNotReady) {
			RX_HIP(hipStreamSynchronize(st));
		} else {
			RX_HIP(q);
		}
	}
	if (h->res_pending) {   // the merge in front of the fusion has ended too: settle its state without another wait
		h->res_pending = false;
		uint32_t mh[4] = {0, 0, 0, 0};
		RX_HIP(hipMemcpy(mh, h->d_out.ptr, sizeof(mh), hipMemcpyDeviceToHost));
		float ms = 0.f;
		if (h->ev_a && hipEventElapsedTime(&ms, h->ev_a, h->ev_b) == hipSuccess) h->stat_ms += ms;
		RX_CHECK(mh[1] == 0, RXGPU_ERR_DEVICE, "rxgpu_hybrid_fuse_resident: ordered look-back timed out on the device");
		h->clean_dirty = false;
	}
	{
		float fms = 0.f;
		if (hipEventElapsedTime(&fms, h->ev_fa, h->ev_fb) == hipSuccess) {
			h->fuse_ms += fms;
			h->fuse_calls += 1;
		}
		if (h->prep_timed && hipEventElapsedTime(&fms, h->ev_pa, h->ev_pb) == hipSuccess) h->prep_ms += fms;
		h->prep_timed = false;
	}
	if (stamps) {
		const unsigned long long* raw = reinterpret_cast<const unsigned long long*>(hp + 64);
		for (int k2 = 1; k2 < 8; ++k2) h->fuse_stamps[k2] += raw[k2] >= raw[0] ? double(raw[k2] - raw[0]) * 0.01 : 0.0;   // 100 MHz -> us
	}
	const uint32_t* hdr = reinterpret_cast<const uint32_t*>(hp);
	const uint64_t n = hdr[0];
	RX_CHECK(n <= out_cap, RXGPU_ERR_DEVICE, "rxgpu_hybrid_fuse_resident: corrupt result header");
	if (n) {
		std::memcpy(out_ids, hp + o_ids, n * 4);
		std::memcpy(out_ranks, hp + o_ranks, n * 4);
	}
	*out_n = n;
	if (out_flags) *out_flags = hdr[1];
	return RXGPU_OK;
}

int rxgpu_hybrid_read_stats(rxgpu_ft_index* h, uint64_t* calls, double* kernel_ms, double* prepare_ms) {
	RX_CHECK(h && calls && kernel_ms, RXGPU_ERR_PARAMS, "rxgpu_hybrid_read_stats: null argument");
	std::lock_guard<std::mutex> lk(h->mtx);
	*calls = h->fuse_calls;
	*kernel_ms = h->fuse_ms;
	if (prepare_ms) *prepare_ms = h->prep_ms;
	h->prep_ms = 0.0;
	if (std::getenv("RXGPU_FUSE_STAMPS") && h->fuse_calls) {
		std::fprintf(stderr, "[rxgpu fuse stamps] us since kernel start:");
		for (int k = 1; k < 8; ++k) {
			std::fprintf(stderr, " %d:%.1f", k, h->fuse_stamps[k] / double(h->fuse_calls));
			h->fuse_stamps[k] = 0;
		}
		std::fprintf(stderr, "\n");
	}
	h->fuse_calls = 0;
	h->fuse_ms = 0.0;
	return RXGPU_OK;
}

// The same kernel on host arrays (tests, callers whose two halves are already on the host): everything is staged, fused, brought back.
int rxgpu_hybrid_fuse(int device, const rxgpu_hybrid_params* params, int metric, const int32_t* knn_ids, const float* knn_ranks, uint32_t n_knn,
					  const int32_t* ft_ids, const uint8_t* ft_ranks, uint32_t n_ft, int32_t* out_ids, float* out_ranks, uint64_t cap, uint64_t* out_n) {
	RX_CHECK(out_n, RXGPU_ERR_PARAMS, "rxgpu_hybrid_fuse: null argument");
	*out_n = 0;
	if (int rc = check_hybrid_params(params, "rxgpu_hybrid_fuse"); rc) return rc;
	RX_CHECK(n_knn <= uint32_t(rxgpu::kMaxFuseKnn), RXGPU_ERR_PARAMS, "rxgpu_hybrid_fuse: at most 1024 KNN entries");
	RX_CHECK((n_knn == 0 || (knn_ids && knn_ranks)) && (n_ft == 0 || (ft_ids && ft_ranks)), RXGPU_ERR_PARAMS, "rxgpu_hybrid_fuse: null argument");
	RX_CHECK(cap >= uint64_t(n_knn) + n_ft && (cap == 0 || (out_ids && out_ranks)), RXGPU_ERR_OVERFLOW, "rxgpu_hybrid_fuse: output buffers too small");
	int ndev = 0;
	RX_HIP(hipGetDeviceCount(&ndev));
	RX_CHECK(device >= 0 && device < ndev, RXGPU_ERR_PARAMS, "rxgpu_hybrid_fuse: no such device");
	DevGuard dg(device);
	const size_t nf = std::max<uint32_t>(n_ft, 1), nk = std::max<uint32_t>(n_knn, 1), no = size_t(n_ft) + n_knn + 1;
	Carver cv;
	const size_t o_fid = cv.take(nf * 4), o_fr = cv.take(nf), o_kid = cv.take(nk * 4), o_kr = cv.take(nk * 4), o_key = cv.take(2 * nf * 4),
				 o_cls = cv.take(2 * nf * 2), o_hdr = cv.take(16), o_oid = cv.take(no * 4), o_or = cv.take(no * 4),
				 o_state = cv.take(sizeof(rxgpu::HybridFuseState));
	rxgpu_devbuf buf;
	if (int rc = buf.ensure(cv.off); rc) return rc;
	struct Rel {
		rxgpu_devbuf& b;
		~Rel() { b.release(); }
	} rel{buf};
	char* d = static_cast<char*>(buf.ptr);
	if (n_ft) {
		RX_HIP(hipMemcpy(d + o_fid, ft_ids, size_t(n_ft) * 4, hipMemcpyHostToDevice));
		RX_HIP(hipMemcpy(d + o_fr, ft_ranks, n_ft, hipMemcpyHostToDevice));
	}
	if (n_knn) {
		RX_HIP(hipMemcpy(d + o_kid, knn_ids, size_t(n_knn) * 4, hipMemcpyHostToDevice));
		RX_HIP(hipMemcpy(d + o_kr, knn_ranks, size_t(n_knn) * 4, hipMemcpyHostToDevice));
	}
	rxgpu::HybridFuseArgs a{};
	a.ft_doc = reinterpret_cast<const uint32_t*>(d + o_fid);
	a.ft_rank_u8 = reinterpret_cast<const uint8_t*>(d + o_fr);
	a.ft_n = n_ft;
	a.ft_cap = uint32_t(nf);
	a.knn_dist = reinterpret_cast<const float*>(d + o_kr);   // ranks as the planner holds them: no sign change
	a.knn_row = reinterpret_cast<const uint32_t*>(d + o_kid);
	a.knn_n = n_knn;
	a.k = n_knn;
	a.knn_negate = 0;
	fill_reranker(a, params, metric);
	a.out_header = reinterpret_cast<uint32_t*>(d + o_hdr);
	a.out_ids = reinterpret_cast<int32_t*>(d + o_oid);
	a.out_ranks = reinterpret_cast<float*>(d + o_or);
	a.scratch_key = reinterpret_cast<uint32_t*>(d + o_key);
	a.scratch_cls = reinterpret_cast<uint16_t*>(d + o_cls);
	a.state = reinterpret_cast<rxgpu::HybridFuseState*>(d + o_state);
	RX_HIP(rxgpu::launch_hybrid_prepare(a, nullptr));
	RX_HIP(rxgpu::launch_hybrid_join(a, nullptr));
	RX_HIP(hipStreamSynchronize(nullptr));   // (the two launches above; not a device-wide wait — resident search kernels may be alive)
	uint32_t hdr[4];
	RX_HIP(hipMemcpy(hdr, d + o_hdr, sizeof(hdr), hipMemcpyDeviceToHost));
	const uint64_t n = hdr[0];
	RX_CHECK(n <= uint64_t(n_knn) + n_ft, RXGPU_ERR_DEVICE, "rxgpu_hybrid_fuse: corrupt result header");
	if (n) {
		RX_HIP(hipMemcpy(out_ids, d + o_oid, n * 4, hipMemcpyDeviceToHost));
		RX_HIP(hipMemcpy(out_ranks, d + o_or, n * 4, hipMemcpyDeviceToHost));
	}
	*out_n = n;
	return RXGPU_OK;
}

int rxgpu_ft_read_stats(rxgpu_ft_index* h, uint64_t* postings, double* kernel_ms) {
	RX_CHECK(h && postings && kernel_ms, RXGPU_ERR_PARAMS, "rxgpu_ft_read_stats: null argument");
	std::lock_guard<std::mutex> lk(h->mtx);
	{   // the lanes' merges count too
		std::vector<rxgpu_ft_index*> have;
		{
			std::lock_guard<std::mutex> g(h->lanes_mtx);
			for (auto& l : h->lanes) have.push_back(l.get());
		}
		for (rxgpu_ft_index* l : have) {
			std::lock_guard<std::mutex> ll(l->mtx);
			h->stat_postings += l->stat_postings;
			h->stat_ms += l->stat_ms;
			for (int k = 0; k < 6; ++k) h->trace_us[k] += l->trace_us[k];
			for (int k = 0; k < 64; ++k) h->stamps[k] += l->stamps[k];
			l->stat_postings = 0;
			l->stat_ms = 0.0;
			for (double& v : l->trace_us) v = 0;
			for (double& v : l->stamps) v = 0;
		}
	}
	*postings = h->stat_postings;
	*kernel_ms = h->stat_ms;
	if (const char* e = std::getenv("RXGPU_FT_TRACE"); e && e[0] == '1' && h->trace_us[5] > 0) {
		const double m = h->trace_us[5];
		std::fprintf(stderr, "[rxgpu ft trace] per merge (us): plan %.1f  stage+upload %.1f  launches %.1f  wait+download %.1f  unpack %.1f  (kernels %.1f)\n",
					 h->trace_us[0] / m, h->trace_us[1] / m, h->trace_us[2] / m, h->trace_us[3] / m, h->trace_us[4] / m, h->stat_ms * 1e3 / m);
		if (std::getenv("RXGPU_FT_STAMPS")) {
			auto line = [&](const char* name, int a, int b) {
				std::fprintf(stderr, "[rxgpu ft stamps] %-12s", name);
				for (int k = a; k < b; ++k) std::fprintf(stderr, " %d:%.1f", k, h->stamps[k] / m);
				std::fprintf(stderr, "\n");
			};
			line("ft_ranges", 0, 16);
			line("ft_rank_all", 16, 19);
			line("ft_adders", 24, 32);
			line("ft_finish", 32, 42);
			for (double& v : h->stamps) v = 0;
		}
		for (double& v : h->trace_us) v = 0;
	}
	h->stat_postings = 0;
	h->stat_ms = 0.0;
	return RXGPU_OK;
}

}  // extern "C"
