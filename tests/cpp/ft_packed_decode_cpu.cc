// The device decoder's source (reindexer_amd/csrc/ft_packed_decode.h) compiled for the host: tests/test_ft_packed_decode.py runs it on
// the CPU against PositionPostings::AppendPacked (pinned to the reference's packer) and the entries / range index the C-ABI derives on
// the host.  Test infrastructure only — nothing in the product links this.
#include <cstdint>
#include <cstring>

#include "ft_packed_decode.h"

extern "C" {

// counting pass: returns the status, fills counts[4] = {n, npos, nent, last_doc}
uint32_t ftpk_count(const uint8_t* data, uint64_t len, uint64_t afp, uint32_t num_fields, uint32_t range_docs, uint32_t* counts) {
	const rxgpu::FtPackedCounts c = rxgpu::ft_decode_packed(data, len, afp, num_fields, range_docs, rxgpu::FtPackedOut{});
	counts[0] = c.n;
	counts[1] = c.npos;
	counts[2] = c.nent;
	counts[3] = c.last_doc;
	return c.status;
}

uint32_t ftpk_write(const uint8_t* data, uint64_t len, uint64_t afp, uint32_t num_fields, uint32_t range_docs, uint32_t* doc, uint32_t* pos_off, uint64_t* fpos,
					uint32_t* ent_off, uint8_t* ent_field, uint32_t* ent_tf, uint32_t* ent_first_pos, uint32_t* range_off, uint32_t n_ranges, uint32_t* counts) {
	rxgpu::FtPackedOut o;
	o.doc = doc;
	o.pos_off = pos_off;
	o.fpos = fpos;
	o.ent_off = ent_off;
	o.ent_field = ent_field;
	o.ent_tf = ent_tf;
	o.ent_first_pos = ent_first_pos;
	o.range_off = range_off;
	o.n_ranges = n_ranges;
	const rxgpu::FtPackedCounts c = rxgpu::ft_decode_packed(data, len, afp, num_fields, range_docs, o);
	counts[0] = c.n;
	counts[1] = c.npos;
	counts[2] = c.nent;
	counts[3] = c.last_doc;
	return c.status;
}
}
