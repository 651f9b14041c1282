// PackedIdRelVec streams decoded on the device: the dictionary's posting lists arrive as the reference keeps them
// (cpp_src/core/ft/idrelset.h:155-280, one varint stream per word) in ONE upload, and come out as the flat arrays of the merge kernels —
// instead of the host flattening every word (PositionPostings::AppendPacked) and uploading eight arrays per word.
//
// Two launches: count (postings / positions / entries, validation), then — after the host has laid the words out in one pool — write.
// A stream has no sync points (element lengths depend on flags inside the elements, ids and fields are delta-coded against the element
// before), so the ELEMENT walk of one list is serial; everything around it is not:
//
//   ft_packed_wave<WRITE>   one WAVEFRONT per word.  The stream is consumed in windows of 256 bytes: a coalesced load (4 bytes per lane),
//                           varint ends by ballot, every lane assembles the (<= 4) varints that end in its dword — values + byte offsets
//                           go to LDS in stream order; the walk (IdRelType::unpack / unpackWithoutArrayIdxs, idrelset.cc:74-139, 192-235,
//                           restated as a state machine with ONE consume point) then runs wave-uniform over those values — the scalar unit
//                           does the bit work, no lane diverges, no byte is fetched from memory inside the walk; documents, positions,
//                           (field, tf, first position) entries are staged in LDS and flushed 64 / 128 at a time by all lanes (coalesced),
//                           the range index is filled 64 entries per store.
//                           The COUNT pass walks a word's whole stream (one wavefront) and leaves a CHECKPOINT — byte offset of an element
//                           start + the decoder state in front of it — in every 1 KB piece; the WRITE pass then runs one wavefront per
//                           PIECE, so a long list is written by as many wavefronts as it has pieces (its counting stays one serial walk).
//   ft_packed_count / _write (kept: the one-thread-per-word form of round 2, ft_packed_decode.h shared with the host build that is checked
//                           against the reference packer) serve as the cross-check of the wave kernels in tests/test_gpu_ft_packed.py.
//
// What bounded the thread-per-word form: every byte of a list was a dependent global load of ONE lane (~0.4 us per byte:
// 13.6 ms for the 33 KB lists of the first wavefront), 64 lists of a wavefront in 64 different decoder states, single-lane 4-byte stores.
// Here a list costs ~50 scalar cycles per varint whatever its neighbours do, and the chip runs 4096 lists at a time.
// The COUNT walk stays serial per list: a 1 MB list takes ~20 ms on its wavefront.  The C-ABI entry decodes whatever it is given on the
// device; a caller that wants very long lists decoded on the host says so itself (GpuFtMerger::SetWordsPacked's hostDecodeFromBytes).
#include "ft_packed_decode.h"
#include "rxgpu_internal.h"

namespace rxgpu {

__global__ __launch_bounds__(256) void ft_packed_count(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords,
														 uint32_t num_fields, FtPackedCounts* counts) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords) return;
	const uint64_t b0 = byte_off[2 * w], b1 = byte_off[2 * w + 1];   // (start, end) of the word's stream: the streams lie where the caller packed them
	counts[w] = ft_decode_packed(bytes + b0, b1 - b0, array_found_pos[w], num_fields, kFtRangeDocs, FtPackedOut{});
}

__global__ __launch_bounds__(256) void ft_packed_write(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords,
														 uint32_t num_fields, const FtPackedOut* outs, FtPackedCounts* counts) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords) return;
	const uint64_t b0 = byte_off[2 * w], b1 = byte_off[2 * w + 1];   // (start, end) of the word's stream: the streams lie where the caller packed them
	if (!outs[w].doc) return;   // an empty word
	counts[w] = ft_decode_packed(bytes + b0, b1 - b0, array_found_pos[w], num_fields, kFtRangeDocs, outs[w]);
}

// --------------------------------------------------------------------------------------------------------------- one wavefront per word
namespace {

constexpr int kWinBytes = 256;   // stream bytes per window: 4 per lane
constexpr int kOutPost = 64;     // postings staged before a flush
constexpr int kOutPos = 128;     // positions / entries staged before a flush

struct PwShared {
	uint32_t val[kWinBytes];      // varints of the window in stream order
	uint32_t boff[kWinBytes];     // their first byte, relative to the window
	uint32_t o_doc[kOutPost], o_pos_off[kOutPost], o_ent_off[kOutPost];
	uint64_t o_fpos[kOutPos];
	uint32_t o_ent_tf[kOutPos], o_ent_first[kOutPos];
	uint8_t o_ent_field[kOutPos];
};

enum PwState : uint32_t { kSId, kSHead, kSField, kSArr, kSSize, kSNext, kSDf, kSA };

}  // namespace

// COUNT (WRITE = false): one wavefront per word over the whole stream; leaves a checkpoint at the first element that starts in every
// kFtPackedSegBytes piece.  WRITE: one wavefront per PIECE (seg_word[] names its word, seg_first[word] the word's first piece): it starts
// from its checkpoint and stops where the next one starts — a long list is written by as many wavefronts as it has pieces.
template <bool WRITE>
__global__ __launch_bounds__(64) void ft_packed_wave(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ byte_off,
													  const uint64_t* __restrict__ array_found_pos, uint32_t nwords, uint32_t num_fields,
													  const FtPackedOut* __restrict__ outs, FtPackedCounts* __restrict__ counts,
													  const uint32_t* __restrict__ seg_word, const uint32_t* __restrict__ seg_first,
													  FtPackedCheckpoint* __restrict__ cps, uint32_t first_word) {
	__shared__ PwShared s;
	const uint32_t w = WRITE ? seg_word[blockIdx.x] : blockIdx.x + first_word;   // the count pass may be launched chunk by chunk (behind each chunk's upload)
	if (w >= nwords) return;
	const int lane = threadIdx.x;
	const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
	const uint64_t b0 = byte_off[2 * w], full_len = byte_off[2 * w + 1] - b0;   // (start, end) pairs in launch order
	const uint8_t* __restrict__ data = bytes + b0;
	const uint64_t afp = array_found_pos[w];
	const uint32_t cp0 = seg_first[w], ncp = seg_first[w + 1] - cp0;   // the word's pieces
	FtPackedOut out{};
	uint64_t len = full_len;       // where this wavefront stops
	FtPackedCheckpoint start{};    // ... and what it starts from (all zero: the head of the stream)
	if (WRITE) {
		out = outs[w];
		if (!out.doc) return;   // an empty word
		const uint32_t piece = blockIdx.x - cp0;
		if (piece) {
			start = cps[blockIdx.x];
			if (start.byte_off == ~0ull) return;   // no element starts in this piece: the wavefront in front of it carries on through it
		}
		for (uint32_t nx = piece + 1; nx < ncp; ++nx) {   // the next piece in which an element starts
			const uint64_t bo = cps[cp0 + nx].byte_off;
			if (bo != ~0ull) {
				len = bo;
				break;
			}
		}
	}

	// walk state (wave-uniform)
	uint32_t status = kFtPackedOk;
	uint32_t state = kSId;
	uint64_t n = start.n, npos = start.npos, nent = start.nent;
	uint64_t post_flushed = n, pos_flushed = npos, ent_flushed = nent;   // what has left the LDS staging
	uint32_t last_id = start.last_id, last_field = start.last_field;
	uint32_t next_range = WRITE ? min(start.next_range, out.n_ranges) : 0;   // counting: unclamped (largest document / range + 1 so far)
	uint32_t next_cp = 1;   // counting: the next piece that waits for its checkpoint
	bool first = n == 0;
	// the element in flight
	bool with_arrays = false, same_field = false, need_a = false;
	uint32_t id = 0, pos = 0, field = 0, array_idx = 0, size = 1, first_field = 0, pi = 0, pending = 0;
	bool arr_zero = true, size_is_1 = true;
	uint32_t run_field = 0, run_tf = 0, run_first = 0;

	// (No lambdas around the walk: closures that capture its state by reference made the compiler keep that state in scratch memory —
	// one global-memory round trip per varint.  The staging flushes are macros, the element logic sits behind flags in the loop body.)
#define PW_FLUSH_POSTINGS()                                                  \
	do {                                                                     \
		const uint32_t cnt_ = uint32_t(n - post_flushed);                    \
		__syncthreads();                                                     \
		if (uint32_t(lane) < cnt_) {                                         \
			out.doc[post_flushed + lane] = s.o_doc[lane];                    \
			out.pos_off[post_flushed + lane] = s.o_pos_off[lane];            \
			out.ent_off[post_flushed + lane] = s.o_ent_off[lane];            \
		}                                                                    \
		__syncthreads();                                                     \
		post_flushed = n;                                                    \
	} while (0)
#define PW_FLUSH_POSITIONS()                                                                        \
	do {                                                                                            \
		const uint32_t cnt_ = uint32_t(npos - pos_flushed);                                         \
		__syncthreads();                                                                            \
		for (uint32_t i_ = lane; i_ < cnt_; i_ += 64) out.fpos[pos_flushed + i_] = s.o_fpos[i_];   \
		__syncthreads();                                                                            \
		pos_flushed = npos;                                                                         \
	} while (0)
#define PW_FLUSH_ENTRIES()                                                   \
	do {                                                                     \
		const uint32_t cnt_ = uint32_t(nent - ent_flushed);                  \
		__syncthreads();                                                     \
		for (uint32_t i_ = lane; i_ < cnt_; i_ += 64) {                      \
			out.ent_field[ent_flushed + i_] = s.o_ent_field[i_];             \
			out.ent_tf[ent_flushed + i_] = s.o_ent_tf[i_];                   \
			out.ent_first_pos[ent_flushed + i_] = s.o_ent_first[i_];         \
		}                                                                    \
		__syncthreads();                                                     \
		ent_flushed = nent;                                                  \
	} while (0)
	// the run of positions of one field ends: one (field, tf, first position) entry
#define PW_END_RUN()                                                         \
	do {                                                                     \
		if (WRITE) {                                                         \
			if (nent - ent_flushed == kOutPos) PW_FLUSH_ENTRIES();           \
			if (lane == 0) {                                                 \
				const uint32_t at_ = uint32_t(nent - ent_flushed);           \
				s.o_ent_field[at_] = uint8_t(run_field);                     \
				s.o_ent_tf[at_] = run_tf;                                    \
				s.o_ent_first[at_] = run_first;                              \
			}                                                                \
		}                                                                    \
		++nent;                                                              \
		run_tf = 0;                                                          \
	} while (0)

	uint64_t wpos = start.byte_off;   // first byte of the window; always a varint boundary
	while (wpos < len && status == kFtPackedOk) {
		// ---- the window: 4 bytes per lane, varint ends by ballot, values + byte offsets to LDS in stream order
		const uint32_t wn = uint32_t(min<uint64_t>(kWinBytes, len - wpos));
		uint32_t d = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint32_t p = 4 * lane + j;
			d |= uint32_t(data[wpos + (p < wn ? p : 0)]) << (8 * j);
			if (p >= wn) d = (d & ~(0xFFu << (8 * j))) | (0x80u << (8 * j));   // past the end: a continuation that never ends
		}
		const uint32_t up = __shfl_up(d, 1, 64);   // every lane takes part in the exchange: a lane that sits out would be read as zero
		const uint32_t prev = lane ? up : 0u;      // in front of the window: nothing (it starts on a varint boundary)
		const uint64_t q = (uint64_t(d) << 32) | prev;
		uint32_t ends = 0;   // bit j: byte j of this lane ends a varint
#pragma unroll
		for (int j = 0; j < 4; ++j) ends |= (4 * lane + j < wn && !((d >> (8 * j)) & 0x80u)) ? (1u << j) : 0u;
		const uint64_t e0 = __ballot(ends & 1u), e1 = __ballot(ends & 2u), e2 = __ballot(ends & 4u), e3 = __ballot(ends & 8u);
		const uint32_t nv = uint32_t(__popcll(e0) + __popcll(e1) + __popcll(e2) + __popcll(e3));
		if (nv == 0) {   // 256 bytes (or the rest of the stream) without a varint end
			status = kFtPackedTruncated;
			break;
		}
		uint32_t idx = uint32_t(__popcll(e0 & lt) + __popcll(e1 & lt) + __popcll(e2 & lt) + __popcll(e3 & lt));
		bool bad = false;
#pragma unroll
		for (int j = 0; j < 4; ++j) {   // five continuation bytes in a row (ending at a byte of the stream): not a uint32 varint
			const int qb = 4 + j;
			bool run5 = 4 * lane + j < int(wn);
			for (int t = 0; t < 5; ++t) run5 = run5 && ((q >> (8 * (qb - t))) & 0x80u);
			bad = bad || run5;
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (!(ends & (1u << j))) continue;
			const int qb = 4 + j;   // this byte inside q
			uint32_t k = 0;         // continuation bytes in front of it (at most 4 in a well-formed uint32)
			while (k < 4 && ((q >> (8 * (qb - 1 - int(k)))) & 0x80u)) ++k;
			uint32_t v = 0;
			for (uint32_t i = 0; i <= k; ++i) {
				const uint32_t b = uint32_t(q >> (8 * (qb - int(k) + int(i)))) & 0xFFu;
				v |= (i == 4 ? b : (b & 0x7Fu)) << (7 * i);   // the 5th byte carries bits 28..31 (tools/varint.h:122-176)
			}
			s.val[idx] = v;
			s.boff[idx] = uint32_t(4 * lane + j) - k;
			++idx;
		}
		if (__ballot(bad)) {
			status = kFtPackedTruncated;
			break;
		}
		// the window ends behind its last complete varint — from the ballots, so that the value is wave-uniform BY CONSTRUCTION (a shuffle
		// reduction is a per-lane value to the compiler: the window position, and with it the whole walk, would be compiled as divergent
		// vector code — ~250 exec-masked instructions per varint instead of a few dozen scalar ones)
		uint32_t last_end = 0;
		if (e0) last_end = max(last_end, uint32_t(63 - __builtin_clzll(e0)) * 4 + 0);
		if (e1) last_end = max(last_end, uint32_t(63 - __builtin_clzll(e1)) * 4 + 1);
		if (e2) last_end = max(last_end, uint32_t(63 - __builtin_clzll(e2)) * 4 + 2);
		if (e3) last_end = max(last_end, uint32_t(63 - __builtin_clzll(e3)) * 4 + 3);
		const uint32_t consumed = last_end + 1;
		if (consumed < wn && wn < uint32_t(kWinBytes)) {   // the stream ends inside a varint
			status = kFtPackedTruncated;
		}
		__syncthreads();

		// ---- the walk: one varint per trip, the next one already requested (LDS latency off the chain); values made scalar
		uint32_t nxt_v = s.val[0], nxt_b = s.boff[0];
		for (uint32_t i = 0; i < nv; ++i) {
			const uint32_t v = __builtin_amdgcn_readfirstlane(nxt_v);
			const uint32_t vb = __builtin_amdgcn_readfirstlane(nxt_b);
			const uint32_t ni = i + 1 < nv ? i + 1 : i;
			nxt_v = s.val[ni];
			nxt_b = s.boff[ni];
			bool ok = true, do_begin = false, do_commit = false, do_emit = false;
			switch (state) {
				case kSId:
					if (!WRITE) {   // an element starts here: if it is the first one at or behind the start of a piece, that piece's checkpoint
						const uint64_t pc = min<uint64_t>(ncp - 1, (wpos + vb) / kFtPackedSegBytes);
						if (pc >= next_cp) {   // (pieces skipped on the way — one element spanned them — keep byte_off = ~0)
							if (lane == 0) {
								FtPackedCheckpoint c;
								c.byte_off = wpos + vb;
								c.last_id = last_id;
								c.last_field = last_field;
								c.n = uint32_t(n);
								c.npos = uint32_t(npos);
								c.nent = uint32_t(nent);
								c.next_range = next_range;
								cps[cp0 + uint32_t(pc)] = c;
							}
							next_cp = uint32_t(pc) + 1;
						}
					}
					with_arrays = wpos + vb >= afp;
					id = v;
					state = kSHead;
					break;
				case kSHead: {
					const bool id_modified = v & 1u, field_is_same = v & 2u;
					size_is_1 = v & 4u;
					arr_zero = with_arrays ? bool(v & 8u) : true;
					pos = v >> (with_arrays ? 4 : 3);
					if (id_modified) id += last_id;
					field = last_field;
					array_idx = 0;
					size = 1;
					if (!field_is_same) {
						state = kSField;
					} else if (!arr_zero) {
						state = kSArr;
					} else if (!size_is_1) {
						state = kSSize;
					} else {
						do_begin = true;
					}
					break;
				}
				case kSField:
					field = v;
					if (!arr_zero) {
						state = kSArr;
					} else if (!size_is_1) {
						state = kSSize;
					} else {
						do_begin = true;
					}
					break;
				case kSArr:
					array_idx = v + 1;
					if (!size_is_1) {
						state = kSSize;
					} else {
						do_begin = true;
					}
					break;
				case kSSize:
					if (v == 0xFFFFFFFFu) {   // size - 1 is stored: this one would wrap to a posting without positions
						status = kFtPackedTooLong;
						ok = false;
					} else {
						size = v + 1;
						do_begin = true;
					}
					break;
				case kSNext: {
					uint32_t next = v;
					same_field = next & 1u;
					if (with_arrays) {
						const bool same_arr = next & 2u;
						next >>= 2;
						if (same_field && same_arr) next += pos;
						need_a = !same_arr;
					} else {
						next >>= 1;
						if (same_field) next += pos;
						need_a = false;
					}
					pending = next;
					if (!same_field) {
						state = kSDf;
					} else if (need_a) {
						state = kSA;
					} else {
						do_commit = true;
					}
					break;
				}
				case kSDf:
					field += v;
					if (need_a) {
						state = kSA;
					} else {
						do_commit = true;
					}
					break;
				default:   // kSA
					array_idx = v + (same_field ? array_idx : 0);
					do_commit = true;
					break;
			}
			if (ok && do_begin) {   // the element's header is complete: its document, then its first position (which has no varint of its own)
				if (!first && id <= last_id) {
					status = kFtPackedDocOrder;
					ok = false;
				} else if (n >= 0xFFFFFFFEull || npos + size >= 0xFFFFFFFEull) {
					status = kFtPackedTooLong;
					ok = false;
				} else {
					if (!WRITE) next_range = max(next_range, id / kFtRangeDocs + 1);
					if (WRITE) {
						// range index: every range that starts at or below this document and has no entry yet begins at this posting
						const uint32_t upto = min(out.n_ranges, id / kFtRangeDocs + 1);
						for (uint32_t r = next_range + lane; r < upto; r += 64) out.range_off[r] = uint32_t(n);
						if (upto > next_range) next_range = upto;
						if (n - post_flushed == kOutPost) PW_FLUSH_POSTINGS();
						if (lane == 0) {
							const uint32_t at = uint32_t(n - post_flushed);
							s.o_doc[at] = id;
							s.o_pos_off[at] = uint32_t(npos);
							s.o_ent_off[at] = uint32_t(nent);
						}
					}
					first_field = field;
					run_field = field;
					run_tf = 0;
					run_first = 0;
					pi = 0;
					do_emit = true;
				}
			}
			if (ok && do_commit) {
				pos = pending;
				do_emit = true;
			}
			if (ok && do_emit) {
				if (field >= num_fields) {
					status = kFtPackedField;
					ok = false;
				} else {
					if (run_tf && field != run_field) PW_END_RUN();
					if (!run_tf) {
						run_field = field;
						run_first = pos & ((1u << 28) - 1u);
					}
					++run_tf;
					if (WRITE) {
						if (npos - pos_flushed == kOutPos) PW_FLUSH_POSITIONS();
						if (lane == 0) s.o_fpos[uint32_t(npos - pos_flushed)] = uint64_t(pos) | (uint64_t(array_idx) << 28) | (uint64_t(field) << 56);
					}
					++npos;
					++pi;
					if (pi < size) {
						state = kSNext;
					} else {   // the element is complete
						PW_END_RUN();
						if (nent >= 0xFFFFFFFEull) {
							status = kFtPackedTooLong;
							ok = false;
						} else {
							++n;
							last_id = id;
							last_field = first_field;
							first = false;
							state = kSId;
						}
					}
				}
			}
			if (!ok) break;
		}
		__syncthreads();   // the window's LDS image is rewritten next
		wpos += consumed;
	}
	if (status == kFtPackedOk && state != kSId) status = kFtPackedTruncated;   // the stream (or the piece) ends inside an element
	const bool last_piece = len == full_len;
	if (WRITE && status == kFtPackedOk) {
		if (n > post_flushed) PW_FLUSH_POSTINGS();
		if (npos > pos_flushed) PW_FLUSH_POSITIONS();
		if (nent > ent_flushed) PW_FLUSH_ENTRIES();
		if (last_piece) {
			for (uint32_t r = next_range + lane; r < out.n_ranges; r += 64) out.range_off[r] = uint32_t(n);
			if (lane == 0) {
				out.pos_off[n] = uint32_t(npos);
				out.ent_off[n] = uint32_t(nent);
			}
		}
	}
	if (lane == 0) {
		if (!WRITE) {
			FtPackedCounts c;
			c.n = uint32_t(n);
			c.npos = uint32_t(npos);
			c.nent = uint32_t(nent);
			c.last_doc = last_id;
			c.status = status;
			counts[w] = c;
		} else {   // the pieces of a word report together: totals from the last one, the worst status from any
			if (last_piece) {
				counts[w].n = uint32_t(n);
				counts[w].npos = uint32_t(npos);
				counts[w].nent = uint32_t(nent);
				counts[w].last_doc = last_id;
			}
			if (status != kFtPackedOk) atomicMax(&counts[w].status, status);
		}
	}
}

hipError_t launch_ft_packed_count(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords, uint32_t num_fields,
								   FtPackedCounts* counts, const FtPackedSegs* segs, hipStream_t st, uint32_t first_word, uint32_t word_count) {
	if (!nwords) return hipSuccess;
	if (segs) {   // words [first_word, first_word + word_count) of the nwords the arrays describe
		if (!word_count) return hipSuccess;
		hipLaunchKernelGGL(ft_packed_wave<false>, dim3(word_count), dim3(64), 0, st, bytes, byte_off, array_found_pos, nwords, num_fields, nullptr, counts,
						   segs->seg_word, segs->seg_first, segs->cps, first_word);
	} else {
		hipLaunchKernelGGL(ft_packed_count, dim3((nwords + 255) / 256), dim3(256), 0, st, bytes, byte_off, array_found_pos, nwords, num_fields, counts);
	}
	return hipGetLastError();
}

hipError_t launch_ft_packed_write(const uint8_t* bytes, const uint64_t* byte_off, const uint64_t* array_found_pos, uint32_t nwords, uint32_t num_fields,
								   const FtPackedOut* outs, FtPackedCounts* counts, const FtPackedSegs* segs, hipStream_t st) {
	if (!nwords) return hipSuccess;
	if (segs) {
		hipLaunchKernelGGL(ft_packed_wave<true>, dim3(segs->nsegs), dim3(64), 0, st, bytes, byte_off, array_found_pos, nwords, num_fields, outs, counts,
						   segs->seg_word, segs->seg_first, segs->cps, 0u);
	} else {
		hipLaunchKernelGGL(ft_packed_write, dim3((nwords + 255) / 256), dim3(256), 0, st, bytes, byte_off, array_found_pos, nwords, num_fields, outs, counts);
	}
	return hipGetLastError();
}

}  // namespace rxgpu
