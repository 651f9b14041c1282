"""Test-side writer of PackedIdRelVec byte streams (what PackedIdRelVec::insert_back produces, cpp_src/core/ft/idrelset.h:232-263 with
IdRelType::pack / packWithoutArrayIdxs, idrelset.cc:8-72, 141-190): elements are written without array indexes until the first posting that
carries one, from there on with them; `afp` = the byte offset of that element (the stream length if there is none).

Only a generator of inputs: tests/test_ft_packed_decode.py first checks it against the committed streams of the reference's own packer
(tests/golden/ft.npz) and, where oracle/_ref is present, against the live packer, byte for byte."""
import numpy as np

POS_MASK = (1 << 28) - 1


def _varint(out: bytearray, v: int) -> None:
    v &= 0xFFFFFFFF
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)


def pack_postings(doc, pos_off, fpos):
    out = bytearray()
    afp = None
    last_id, last_field = 0, 0
    doc = [int(d) for d in doc]
    pos_off = [int(x) for x in pos_off]
    words = [int(w) for w in fpos]
    for i, d in enumerate(doc):
        ps = [(w & POS_MASK, (w >> 28) & POS_MASK, w >> 56) for w in words[pos_off[i]:pos_off[i + 1]]]   # (pos, arrayIdx, field)
        assert ps
        if afp is None and any(a > 0 for _, a, _ in ps):
            afp = len(out)
        with_arrays = afp is not None
        id_modified = d >= last_id
        _varint(out, d - last_id if id_modified else d)
        pos, arr, field = ps[0]
        same_field, size1 = field == last_field, len(ps) == 1
        if with_arrays:
            arr0 = arr == 0
            _varint(out, (pos << 4) | int(id_modified) | (int(same_field) << 1) | (int(size1) << 2) | (int(arr0) << 3))
            if not same_field:
                _varint(out, field)
            if not arr0:
                _varint(out, arr - 1)
        else:
            _varint(out, (pos << 3) | int(id_modified) | (int(same_field) << 1) | (int(size1) << 2))
            if not same_field:
                _varint(out, field)
        if not size1:
            _varint(out, len(ps) - 1)
        first_field = field
        for npos, narr, nfield in ps[1:]:
            sf = nfield == field
            if with_arrays:
                sa = narr == arr
                shift = npos - pos if (sf and sa) else npos
                _varint(out, (shift << 2) | int(sf) | (int(sa) << 1))
                if not sf:
                    _varint(out, nfield - field)
                if not sa:
                    _varint(out, narr - arr if sf else narr)
            else:
                shift = npos - pos if sf else npos
                _varint(out, (shift << 1) | int(sf))
                if not sf:
                    _varint(out, nfield - field)
            pos, arr, field = npos, narr, nfield
        last_id, last_field = d, first_field
    data = np.frombuffer(bytes(out), np.uint8).copy()
    return data, (len(out) if afp is None else afp)


def flat_entries(doc, pos_off, fpos, range_docs=8192):
    """The (field, tf, first position) entries and the range index rxgpu_ft_set_word_positions / rxgpu_ft_set_word derive on the host."""
    ent_off, ent_field, ent_tf, ent_first = [0], [], [], []
    for i in range(len(doc)):
        a = int(pos_off[i])
        e = int(pos_off[i + 1])
        while a < e:
            f = int(fpos[a]) >> 56
            b = a + 1
            while b < e and (int(fpos[b]) >> 56) == f:
                b += 1
            ent_field.append(f)
            ent_tf.append(b - a)
            ent_first.append(int(fpos[a]) & POS_MASK)
            a = b
        ent_off.append(len(ent_field))
    n = len(doc)
    n_ranges = (int(doc[-1]) // range_docs + 2) if n else 0
    ro, i = [], 0
    for k in range(n_ranges):
        while i < n and int(doc[i]) < k * range_docs:
            i += 1
        ro.append(i)
    return (np.array(ent_off, np.uint32), np.array(ent_field, np.uint8), np.array(ent_tf, np.uint32), np.array(ent_first, np.uint32),
            np.array(ro, np.uint32))
