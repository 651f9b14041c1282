// Host side of the resident HNSW search kernel (hnsw_server_kernel, hnsw_server.hip; protocol: HnswServer, knn_kernels.hip.h).
//
// The reference's planner issues ONE query per HnswIndexBase::select (hnsw_index.cc:159-288 -> hnswalg.h:1988-2012) from as many threads as
// it has connections (gtests/tests/unit/float_vector_index.cc:258-294 runs 16).  As a launch per call that path was bound by everything
// around the search: 9.9 k q/s at T = 16 over 10M x 768 against 18.6 k for the reference's 16 cores.  Here a call claims a slot of a mailbox in
// pinned host memory, stores its query and a sequence number, and polls the slot's answer; the kernel that serves the mailbox is launched by
// whichever caller finds none alive and ends by itself (stop word / idle / lifetime), so nothing on the device ever waits for the host.
// Measured at the end of round 6 (profiles/rd6zz_bench_full.json): T = 16 / 64 / 256 -> 21.9 k / 82.4 k / 175 k q/s over 10M x 768, every search
// 0.67 - 0.69 ms on the device whatever T; the reference's 16 cores 19.1 k q/s.
//
// What is NOT served here (the caller takes the ordinary launches): batches, SQ8 graphs, ef above 256 (224 with deleted nodes; an index has a
// mailbox per list size: ef <= 128 / 96 and above), embedding sizes without a fixed-dimension distance batch, a profiled index, every RXGPU_HNSW_* A/B hook that names a
// kernel form.  A search that comes back flagged (equal keys that the in-kernel restart could not settle, a visited set half full) is
// answered by the launches' re-run tiers inside the same call.
#include <immintrin.h>
#include <sched.h>
#include <sys/prctl.h>
#include <time.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <set>

#include "../../include/rxgpu.h"
#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

namespace rxgpu {

namespace {
constexpr uint32_t kServerEfCapOf[2] = {128u, 256u};   // the two classes of resident kernel (hnsw_server.hip): lists of two / four entries a lane
constexpr uint32_t kServerWideVisLog2 = 15;              // class 1: a slot's visited hash set in HBM, 2^15 words (as a launch of that ef gets)
constexpr uint32_t kServerRestartCap = 600; // heap area of a search that starts over on the reference's heaps (as a team launch gets)
constexpr uint32_t kServerVisLog2 = 14;     // 16384-word hash set in LDS (64 KB of the CU's 160): a search may mark 8192 nodes — at 10M x 768, ef = 128, 0.7 % of
                                            // the searches marked more than the 4096 a 32 KB set allows and went back to the launches

std::mutex g_servers_mtx;
std::set<HnswServerState*> g_servers;

inline uint32_t load_acq(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void store_rel(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
}  // namespace

struct HnswServerState {
	std::mutex mtx;   // launches, quiesce, teardown
	int device = 0;
	hipStream_t stream = nullptr;
	char* host = nullptr;               // the mailbox
	char* dev_view = nullptr;           // ... as the device addresses it
	unsigned long long* d_words = nullptr;
	uint32_t slots = 0, dim = 0, cls = 0, kcap = 128;
	uint32_t* d_visited = nullptr;      // class 1: [slots][2^kServerWideVisLog2] words
	size_t o_took = 0, o_post = 0, o_done = 0, o_req = 0, o_stop = 0, o_leaving = 0, o_count = 0, o_query = 0, o_dist = 0, o_row = 0, bytes = 0;
	std::atomic<uint64_t> free_mask[4];
	std::vector<uint32_t> seq;          // per slot; touched by the slot's holder only
	std::atomic<uint32_t> launched{0};  // generation number of the newest launch
	std::atomic<bool> broken{false};
	std::atomic<bool> maybe_alive{false};   // a generation was launched since the stream was last seen idle
	std::atomic<uint64_t> served{0}, generations{0};
	std::atomic<uint32_t> in_flight{0};   // requests posted and not yet answered
	std::atomic<uint64_t> device_ticks{0}, caller_us{0};   // sums over the answered requests: the search on the device (100 MHz ticks) / post -> answer seen by the caller
	std::atomic<uint32_t> expect_us{0};   // running estimate of a request's duration (see the wait in hnsw_server_search)
	unsigned long long idle_ticks = 0, life_ticks = 0;
	bool spec = false, nbl = false;

	uint32_t* post() const { return reinterpret_cast<uint32_t*>(host + o_post); }
	uint32_t* done() const { return reinterpret_cast<uint32_t*>(host + o_done); }
	uint32_t* req() const { return reinterpret_cast<uint32_t*>(host + o_req); }
	uint32_t* stop() const { return reinterpret_cast<uint32_t*>(host + o_stop); }
	uint32_t* leaving() const { return reinterpret_cast<uint32_t*>(host + o_leaving); }
};

static void server_free(HnswServerState* st) {
	if (!st) return;
	if (st->stream) (void)hipStreamDestroy(st->stream);
	if (st->host) (void)hipHostFree(st->host);
	if (st->d_words) (void)hipFree(st->d_words);
	if (st->d_visited) (void)hipFree(st->d_visited);
	delete st;
}

// (under h->mtx) the index's mailbox, made at the first single query
static HnswServerState* server_create(rxgpu_index* h, uint32_t cls, uint32_t slots, uint32_t idle_us, uint32_t life_ms, bool spec, bool nbl) {
	auto* st = new HnswServerState();
	st->cls = cls;
	st->kcap = kServerEfCapOf[cls];
	st->spec = spec;
	st->nbl = nbl;
	st->device = h->device;
	st->slots = slots;
	st->dim = h->dim;
	st->idle_ticks = uint64_t(idle_us) * 100ull;
	st->life_ticks = uint64_t(life_ms) * 100000ull;
	auto take = [&](size_t bytes) {
		const size_t at = st->bytes;
		st->bytes += (bytes + 255) & ~size_t(255);
		return at;
	};
	st->o_post = take(size_t(slots) * 4);
	st->o_done = take(size_t(slots) * 4);
	st->o_took = take(size_t(slots) * 4);
	st->o_req = take(size_t(slots) * 8);
	st->o_stop = take(4);
	st->o_leaving = take(4);
	st->o_count = take(size_t(slots) * 4);
	st->o_query = take(size_t(slots) * h->dim * 4);
	st->o_dist = take(size_t(slots) * st->kcap * 4);
	st->o_row = take(size_t(slots) * st->kcap * 4);
	int least = 0, greatest = 0;
	void* dv = nullptr;
	// a stream of the highest priority: the runtime keeps a pool of hardware queues per priority, so the resident kernel never sits in a
	// queue in front of an ordinary stream's launches (which would wait for it to leave)
	if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
		hipStreamCreateWithPriority(&st->stream, hipStreamNonBlocking, greatest) != hipSuccess ||
		hipHostMalloc(reinterpret_cast<void**>(&st->host), st->bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
		hipHostGetDevicePointer(&dv, st->host, 0) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&st->d_words), 2 * sizeof(unsigned long long)) != hipSuccess ||
		(cls == 1 && hipMalloc(reinterpret_cast<void**>(&st->d_visited), (size_t(slots) << kServerWideVisLog2) * 4) != hipSuccess)) {
		(void)hipGetLastError();
		server_free(st);
		return nullptr;
	}
	st->dev_view = static_cast<char*>(dv);
	std::memset(st->host, 0, st->bytes);
	for (uint32_t w = 0; w < 4; ++w) {
		const uint32_t lo = 64 * w;
		st->free_mask[w].store(slots >= lo + 64 ? ~0ull : (slots > lo ? ((1ull << (slots - lo)) - 1ull) : 0ull));
	}
	st->seq.assign(slots, 0u);
	std::lock_guard<std::mutex> lk(g_servers_mtx);
	g_servers.insert(st);
	return st;
}

// (under st->mtx) one more generation, if none is alive or queued
static int server_launch(rxgpu_index* h, HnswServerState* st) {
	if (load_acq(st->leaving()) != st->launched.load(std::memory_order_acquire)) return RXGPU_OK;   // another caller was first
	if (load_acq(st->stop())) return RXGPU_OK;                                                        // the index is changing
	HnswParams p{};
	p.rows = h->d_rows;
	p.inv_norms = h->d_inv_norms;
	p.links0 = h->d_links0;
	p.upper_off = h->d_upper_off;
	p.upper = h->d_upper;
	p.deleted = h->d_deleted;
	p.n = h->count;
	p.stride = h->stride;
	p.dim = h->dim;
	p.M = h->graph_M;
	p.maxM0 = h->graph_maxM0;
	p.maxlevel = h->graph_maxlevel;
	p.entry = h->graph_entry;
	p.bare = h->graph_deleted == 0;
	p.nq = st->slots;
	p.ef = 1;   // (per request; the launcher checks the class limits against what the host admits)
	p.k = 1;
	if (st->cls == 0) {   // the visited set of a search in the workgroup's LDS
		p.visited = nullptr;
		p.visited_words = 0;
		p.vis_hash_log2 = kServerVisLog2;
		p.vis_lds_log2 = kServerVisLog2;
		p.vis_lds = 1;
	} else {              // ... in the slot's part of an HBM table (64 ef words would not fit the LDS)
		p.visited = st->d_visited;
		p.visited_words = uint64_t(1) << kServerWideVisLog2;
		p.vis_hash_log2 = kServerWideVisLog2;
		p.vis_lds_log2 = 0;
		p.vis_lds = 0;
	}
	p.prefetch_links = 1;
	p.team = 4;
	p.team_max = st->slots;
	p.spec = st->spec ? 1u : 0u;
	p.nbl = st->nbl ? 1u : 0u;
	p.queries = reinterpret_cast<const float*>(st->dev_view + st->o_query);
	p.out_dist = reinterpret_cast<float*>(st->dev_view + st->o_dist);
	p.out_row = reinterpret_cast<uint32_t*>(st->dev_view + st->o_row);
	p.out_count = reinterpret_cast<uint32_t*>(st->dev_view + st->o_count);
	p.stats = h->d_hnsw_stats;
	p.ef_cap = kServerEfCapOf[st->cls];
	p.lds_cand_cap = kServerRestartCap;
	p.sorted = 1;
	HnswServer sv{};
	sv.post = reinterpret_cast<const uint32_t*>(st->dev_view + st->o_post);
	sv.done = reinterpret_cast<uint32_t*>(st->dev_view + st->o_done);
	sv.took = reinterpret_cast<uint32_t*>(st->dev_view + st->o_took);
	sv.req = reinterpret_cast<const uint32_t*>(st->dev_view + st->o_req);
	sv.stop = reinterpret_cast<const uint32_t*>(st->dev_view + st->o_stop);
	sv.leaving = reinterpret_cast<uint32_t*>(st->dev_view + st->o_leaving);
	sv.dev = st->d_words;
	sv.generation = st->launched.load(std::memory_order_relaxed) + 1u;
	sv.kcap = st->kcap;
	sv.idle_ticks = st->idle_ticks;
	sv.life_ticks = st->life_ticks;
	if (hipMemsetAsync(st->d_words, 0, 2 * sizeof(unsigned long long), st->stream) != hipSuccess || !launch_hnsw_server(h->metric, p, sv, st->slots, st->stream) ||
		hipGetLastError() != hipSuccess) {
		(void)hipGetLastError();
		st->broken.store(true);
		set_error("hnsw server: launch failed");
		return RXGPU_ERR_DEVICE;
	}
	if (!st->maybe_alive.exchange(true, std::memory_order_acq_rel)) g_resident_kernels.fetch_add(1, std::memory_order_acq_rel);
	st->launched.store(sv.generation, std::memory_order_release);
	st->generations.fetch_add(1, std::memory_order_relaxed);
	return RXGPU_OK;
}

static void server_quiesce_state(HnswServerState* st) {
	if (!st->maybe_alive.load(std::memory_order_acquire)) return;   // (mutators call this per row: no lock, no driver call when nothing ran)
	std::lock_guard<std::mutex> lk(st->mtx);
	if (!st->maybe_alive.load(std::memory_order_acquire)) return;
	DeviceGuardLite dg(st->device);
	store_rel(st->stop(), 1u);
	(void)hipStreamSynchronize(st->stream);
	store_rel(st->leaving(), st->launched.load());   // (a generation that ended on its wall-clock fallback never wrote it)
	store_rel(st->stop(), 0u);
	st->maybe_alive.store(false, std::memory_order_release);
	if (g_resident_kernels.fetch_sub(1, std::memory_order_acq_rel) == 1) drain_retired();   // (no resident kernel left: what was put aside meanwhile goes now)
}

void hnsw_server_quiesce(rxgpu_index* h) {
	if (!h) return;
	for (HnswServerState* st : h->hnsw_server) {
		if (st) server_quiesce_state(st);
	}
}

void hnsw_servers_pause_device(int device) {
	std::lock_guard<std::mutex> lk(g_servers_mtx);
	for (HnswServerState* st : g_servers) {
		if (st->device == device) server_quiesce_state(st);
	}
}

void hnsw_server_destroy(rxgpu_index* h) {
	if (!h) return;
	for (HnswServerState*& slot : h->hnsw_server) {
		HnswServerState* st = slot;
		if (!st) continue;
		server_quiesce_state(st);
		{
			std::lock_guard<std::mutex> lk(g_servers_mtx);
			g_servers.erase(st);
		}
		slot = nullptr;
		server_free(st);
	}
}

void hnsw_server_times(const rxgpu_index* h, uint64_t* device_us, uint64_t* caller_us) {
	*device_us = *caller_us = 0;
	if (!h) return;
	for (const HnswServerState* st : h->hnsw_server) {
		if (st) {
			*device_us += st->device_ticks.load() / 100u;
			*caller_us += st->caller_us.load();
		}
	}
}

void hnsw_server_counters(const rxgpu_index* h, uint64_t* served, uint64_t* generations) {
	*served = *generations = 0;
	if (!h) return;
	for (const HnswServerState* st : h->hnsw_server) {
		if (st) {
			*served += st->served.load();
			*generations += st->generations.load();
		}
	}
}

// 1: served (out_* hold the result), 0: not taken — the caller takes the launches, 2: taken, but the search came back flagged (equal keys the
// in-kernel restart could not settle, a visited set half full): the launches' re-run tiers answer it; any other value: an RXGPU error code
int hnsw_server_search(rxgpu_index* h, const HnswServerConfig& cfg, const float* query, uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row,
					   uint32_t* out_count) {
	if (h->profiling) return 0;
	if (h->dim != 128 && h->dim != 512 && h->dim != 768) return 0;
	const bool bare = h->graph_deleted == 0;
	if (ef > (bare ? 256u : 224u) || k > ef) return 0;
	const uint32_t cls = ef > (bare ? 128u : 96u) ? 1u : 0u;
	if (k > kServerEfCapOf[cls]) return 0;
	HnswServerState* st = h->hnsw_server[cls];
	if (!st) {
		std::lock_guard<std::mutex> lk(h->mtx);
		if (!h->hnsw_server[cls]) {
			if (h->hnsw_server_failed[cls]) return 0;
			h->hnsw_server[cls] = server_create(h, cls, std::min<uint32_t>(256u, std::max<uint32_t>(1u, cfg.slots)), cfg.idle_us, cfg.life_ms, cfg.spec, cfg.nbl);
			if (!h->hnsw_server[cls]) {
				h->hnsw_server_failed[cls] = true;
				return 0;
			}
		}
		st = h->hnsw_server[cls];
	}
	if (st->broken.load(std::memory_order_relaxed)) return 0;
	// a free slot (none: every workgroup is busy — this query takes a launch of its own)
	int slot = -1;
	for (uint32_t w = 0; w < 4 && slot < 0; ++w) {
		uint64_t m = st->free_mask[w].load(std::memory_order_relaxed);
		while (m) {
			const int b = __builtin_ctzll(m);
			const uint64_t bit = 1ull << b;
			if (st->free_mask[w].fetch_and(~bit, std::memory_order_acquire) & bit) {
				slot = int(64 * w) + b;
				break;
			}
			m = st->free_mask[w].load(std::memory_order_relaxed);
		}
	}
	if (slot < 0) return 0;
	struct Release {
		HnswServerState* st;
		int slot;
		~Release() { st->free_mask[slot >> 6].fetch_or(1ull << (slot & 63), std::memory_order_release); }
	} release{st, slot};
	std::memcpy(st->host + st->o_query + size_t(slot) * st->dim * 4, query, size_t(st->dim) * 4);
	st->req()[2 * slot] = k;
	st->req()[2 * slot + 1] = ef;
	const uint32_t seq = ++st->seq[slot];
	store_rel(st->post() + slot, seq);
	const uint32_t* done = st->done() + slot;
	const auto t0 = std::chrono::steady_clock::now();
	auto waited_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
	// How to wait.  A search lasts some hundred microseconds.  With a few requests in flight the caller spins (then yields between looks):
	// the answer is taken the moment it is there.  With many — T planner threads — spinning callers are T busy cores for nothing, and
	// under a CPU quota (a container with 16 CPUs' worth of a larger host) they are throttled like any other load: 16 spinning threads
	// measured 13.7 k q/s at 10M rows where one thread's 0.67 ms per query allows 24 k.  So from kSpinners requests on, a caller sleeps
	// through most of the expected duration (the running estimate below: the shortest recent answer, creeping up by 1 us per query) and
	// then looks every ~20 us between short sleeps; its CPU time per query is a few microseconds.
	constexpr uint32_t kSpinners = 4;
	const uint32_t in_flight = st->in_flight.fetch_add(1, std::memory_order_relaxed) + 1u;
	struct Leave {
		std::atomic<uint32_t>& n;
		~Leave() { n.fetch_sub(1, std::memory_order_relaxed); }
	} leave{st->in_flight};
	const bool sleeper = in_flight > kSpinners;
	auto nap = [](uint32_t us) {
		thread_local bool slack_set = false;
		if (!slack_set) {   // (the default 50 us of timer slack would make every 20 us nap a 70 us one)
			(void)prctl(PR_SET_TIMERSLACK, 2000UL, 0UL, 0UL, 0UL);
			slack_set = true;
		}
		timespec ts{0, long(us) * 1000L};
		(void)nanosleep(&ts, nullptr);
	};
	auto ensure_alive = [&]() -> int {
		if (load_acq(st->leaving()) == st->launched.load(std::memory_order_acquire) && !load_acq(st->stop())) {   // no generation alive or queued
			std::lock_guard<std::mutex> lk(st->mtx);
			DeviceGuardLite dg(st->device);
			return server_launch(h, st);
		}
		return RXGPU_OK;
	};
	if (int rc = ensure_alive(); rc) return rc;
	if (sleeper) {
		const uint32_t expect = st->expect_us.load(std::memory_order_relaxed);
		if (expect > 150u) nap(expect - 100u);
	}
	double next_check = 50.0;   // us: when to look at the generation again (it may have left between two looks at the answer)
	bool stale_checked = false;
	for (uint32_t it = 0;; ++it) {
		if (load_acq(done) == seq) break;
		if (it < 128u) {
			_mm_pause();
			continue;
		}
		const double w = waited_us();
		if (w >= next_check) {
			next_check = w + 50.0;
			if (int rc = ensure_alive(); rc) return rc;
			if (w > 20000.0 && !stale_checked) {   // far beyond a search: did the generation end without saying so (its wall-clock fallback)?
				stale_checked = true;
				std::lock_guard<std::mutex> lk(st->mtx);
				if (load_acq(st->leaving()) != st->launched.load() && hipStreamQuery(st->stream) == hipSuccess) store_rel(st->leaving(), st->launched.load());
			}
			if (w > 5e6) {
				st->broken.store(true);
				set_error("hnsw server: no answer within 5 s");
				return RXGPU_ERR_DEVICE;
			}
		}
		if (sleeper) {
			nap(15u);
		} else {
			sched_yield();
		}
	}
	st->device_ticks.fetch_add(*reinterpret_cast<const volatile uint32_t*>(st->host + st->o_took + size_t(slot) * 4), std::memory_order_relaxed);
	st->caller_us.fetch_add(uint64_t(std::min(waited_us(), 1e9)), std::memory_order_relaxed);
	{   // the estimate the sleepers use: never above what was just seen, one microsecond up per query (so it follows a growing index)
		const uint32_t took = uint32_t(std::min(waited_us(), 1e6));
		const uint32_t e = st->expect_us.load(std::memory_order_relaxed);
		st->expect_us.store(e == 0u ? took : std::min(e + 1u, took), std::memory_order_relaxed);
	}
	const uint32_t count = *reinterpret_cast<const volatile uint32_t*>(st->host + st->o_count + size_t(slot) * 4);
	if (count == kHnswTie || count == kHnswOverflow || count > k) return 2;   // the search ran and needs the re-run tiers: not a case for a second look at the mailbox
	std::memcpy(out_dist, st->host + st->o_dist + size_t(slot) * st->kcap * 4, size_t(count) * 4);
	std::memcpy(out_row, st->host + st->o_row + size_t(slot) * st->kcap * 4, size_t(count) * 4);
	*out_count = count;
	st->served.fetch_add(1, std::memory_order_relaxed);
	return 1;
}

}  // namespace rxgpu
