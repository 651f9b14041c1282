// Decoder of ONE PackedIdRelVec byte stream (the reference's varint-packed posting list of a dictionary word) into the flat arrays the
// merge kernels read.  Plain C++ that compiles for the device (ft_packed.hip: one thread per word) and for the host (tests/cpp: the
// same code checked on the CPU against the host decoder PositionPostings::AppendPacked, which is pinned to the reference's own packer).
//
// Wire format (cpp_src/core/ft/idrelset.cc:8-139 pack / unpack, idrelset.h:155-280 PackedIdRelVec, tools/varint.h:122-176):
//   element  = [id][head][field?][arrayIdx - 1?][size - 1?] then (size - 1) x ([next][fieldDelta?][arrayIdx?])
//   id       delta against the previous element's id when head bit 0 is set
//   head     pos << 4 | idModified | fieldIsSame << 1 | sizeIs1 << 2 | arrayIdxIsZero << 3; elements stored in front of byte
//            `array_found_pos` use the older layout without array indexes: pos << 3, no bit 3, next = pos << 1 | sameField
//   "same field" refers to the FIRST position's field of the previous element (PackedIdRelVec::state), 0 in front of the first element
//   varints  base 128, little endian, at most 5 bytes for a uint32 (the 5th byte carries bits 28..31)
// Output per posting: the document, its positions as PosType words (pos | arrayIdx << 28 | field << 56, idrelset.h:14-32), and the
// (field, tf, first position) entries calcTermRankImpl groups out of them (phrasemergerimpl.h:24-49) — what rxgpu_ft_set_word_positions
// derives on the host; plus the range index over the ascending documents (first posting with doc >= k * range_docs).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define RX_HD __host__ __device__
#else
#define RX_HD
#endif

namespace rxgpu {

enum FtPackedStatus : uint32_t {
	kFtPackedOk = 0,
	kFtPackedTruncated = 1,    // a varint or an element runs past the end of the stream
	kFtPackedDocOrder = 2,     // document ids do not ascend strictly
	kFtPackedField = 3,        // a field number >= num_fields
	kFtPackedTooLong = 4,      // more than 2^32 - 2 postings / positions / entries in one word
};

struct FtPackedCounts {
	uint32_t n = 0, npos = 0, nent = 0;   // postings, positions, (field, tf, first position) entries
	uint32_t last_doc = 0;
	uint32_t status = kFtPackedOk;
	uint32_t pad[3] = {0, 0, 0};
};

struct FtPackedOut {   // all null for the counting pass
	uint32_t* doc = nullptr;         // [n]
	uint32_t* pos_off = nullptr;     // [n + 1]
	uint64_t* fpos = nullptr;        // [npos]
	uint32_t* ent_off = nullptr;     // [n + 1]
	uint8_t* ent_field = nullptr;    // [nent]
	uint32_t* ent_tf = nullptr;      // [nent]
	uint32_t* ent_first_pos = nullptr;   // [nent]
	uint32_t* range_off = nullptr;   // [n_ranges]
	uint32_t n_ranges = 0;           // last_doc / range_docs + 2
};

// A point of a stream where an element starts, with the decoder state in front of it (ft_packed_wave: the counting pass leaves one per
// kFtPackedSegBytes of stream, the writing pass decodes the pieces between them with one wavefront each).
constexpr uint32_t kFtPackedSegBytes = 1024;
struct FtPackedCheckpoint {
	uint64_t byte_off;   // ~0: no element starts inside this piece (one element spans it)
	uint32_t last_id, last_field;
	uint32_t n, npos, nent;
	uint32_t next_range;   // unclamped: largest (document / range_docs + 1) seen so far
};

RX_HD inline bool ft_packed_varint(const uint8_t*& p, const uint8_t* end, uint32_t& v) {
	v = 0;
	for (unsigned i = 0; i < 5; ++i) {
		if (p == end) return false;
		const uint8_t b = *p++;
		if (i == 4) {
			v |= uint32_t(b) << 28;
			return true;
		}
		v |= uint32_t(b & 0x7f) << (7 * i);
		if (!(b & 0x80)) return true;
	}
	return true;
}

// Decodes the stream [data, data + len).  With out.doc == nullptr nothing is written and the counts come back; with the arrays in place
// (sized by a counting pass) everything is written.  On an error the status is set and decoding stops.
RX_HD inline FtPackedCounts ft_decode_packed(const uint8_t* data, uint64_t len, uint64_t array_found_pos, uint32_t num_fields, uint32_t range_docs,
											 const FtPackedOut& out) {
	FtPackedCounts c;
	const bool write = out.doc != nullptr;
	const uint8_t* p = data;
	const uint8_t* const end = data + len;
	uint32_t last_id = 0, last_field = 0;   // PackedIdRelVec::state (idrelset.h:166-170)
	uint64_t n = 0, npos = 0, nent = 0;
	uint32_t next_range = 0;   // range index entries written so far
	bool first = true;
	while (p != end) {
		const bool with_arrays = uint64_t(p - data) >= array_found_pos;
		uint32_t id, head;
		if (!ft_packed_varint(p, end, id) || !ft_packed_varint(p, end, head)) {
			c.status = kFtPackedTruncated;
			break;
		}
		const bool id_modified = head & 1, field_is_same = head & 2, size_is_1 = head & 4;
		const bool array_idx_is_zero = with_arrays ? bool(head & 8) : true;
		uint32_t pos = head >> (with_arrays ? 4 : 3);
		if (id_modified) id += last_id;
		uint32_t field = last_field, array_idx = 0, size = 1;
		bool ok = true;
		if (!field_is_same) ok = ft_packed_varint(p, end, field);
		if (ok && !array_idx_is_zero) {
			ok = ft_packed_varint(p, end, array_idx);
			array_idx += 1;
		}
		if (ok && !size_is_1) {
			ok = ft_packed_varint(p, end, size);
			if (ok && size == 0xFFFFFFFFu) {   // size - 1 is stored: this one would wrap to a posting without positions
				c.status = kFtPackedTooLong;
				break;
			}
			size += 1;
		}
		if (!ok) {
			c.status = kFtPackedTruncated;
			break;
		}
		if (!first && id <= last_id) {
			c.status = kFtPackedDocOrder;
			break;
		}
		if (n >= 0xFFFFFFFEull || npos + size >= 0xFFFFFFFEull) {
			c.status = kFtPackedTooLong;
			break;
		}
		if (write) {
			while (next_range < out.n_ranges && uint64_t(next_range) * range_docs <= id) out.range_off[next_range++] = uint32_t(n);
			out.doc[n] = id;
			out.pos_off[n] = uint32_t(npos);
			out.ent_off[n] = uint32_t(nent);
		}
		const uint32_t first_field = field;
		// positions; an entry = a run of positions with one field
		uint32_t run_field = field, run_tf = 0, run_first = 0;
		for (uint32_t i = 0; i < size; ++i) {
			if (i) {
				uint32_t next;
				if (!ft_packed_varint(p, end, next)) {
					ok = false;
					break;
				}
				const bool same_field = next & 1;
				if (with_arrays) {
					const bool same_array_idx = next & 2;
					next >>= 2;
					if (same_field && same_array_idx) next += pos;
					if (!same_field) {
						uint32_t df;
						if (!ft_packed_varint(p, end, df)) {
							ok = false;
							break;
						}
						field += df;
					}
					if (!same_array_idx) {
						uint32_t a;
						if (!ft_packed_varint(p, end, a)) {
							ok = false;
							break;
						}
						array_idx = a + (same_field ? array_idx : 0);
					}
				} else {
					next >>= 1;
					if (same_field) {
						next += pos;
					} else {
						uint32_t df;
						if (!ft_packed_varint(p, end, df)) {
							ok = false;
							break;
						}
						field += df;
					}
				}
				pos = next;
			}
			if (field >= num_fields) {
				c.status = kFtPackedField;
				ok = false;
				break;
			}
			if (run_tf && field != run_field) {   // the run of the field before ends
				if (write) {
					out.ent_field[nent] = uint8_t(run_field);
					out.ent_tf[nent] = run_tf;
					out.ent_first_pos[nent] = run_first;
				}
				++nent;
				run_tf = 0;
			}
			if (!run_tf) {
				run_field = field;
				run_first = pos & ((1u << 28) - 1u);
			}
			++run_tf;
			if (write) out.fpos[npos] = uint64_t(pos) | (uint64_t(array_idx) << 28) | (uint64_t(field) << 56);
			++npos;
		}
		if (!ok) {
			if (c.status == kFtPackedOk) c.status = kFtPackedTruncated;
			break;
		}
		if (write) {
			out.ent_field[nent] = uint8_t(run_field);
			out.ent_tf[nent] = run_tf;
			out.ent_first_pos[nent] = run_first;
		}
		++nent;
		if (nent >= 0xFFFFFFFEull) {
			c.status = kFtPackedTooLong;
			break;
		}
		++n;
		last_id = id;
		last_field = first_field;
		first = false;
	}
	if (write && c.status == kFtPackedOk) {
		while (next_range < out.n_ranges) out.range_off[next_range++] = uint32_t(n);
		out.pos_off[n] = uint32_t(npos);
		out.ent_off[n] = uint32_t(nent);
	}
	c.n = uint32_t(n);
	c.npos = uint32_t(npos);
	c.nent = uint32_t(nent);
	c.last_doc = last_id;
	return c;
}

}  // namespace rxgpu
