"""Commit-time side of the ft half: a dictionary's posting lists arrive as PackedIdRelVec byte streams.  Times
  device : GpuFtMerger::SetWordsPacked — one upload of the bytes, decode on the GPU (ft_packed.hip), one pool allocation
  host   : the previous path — PositionPostings::AppendPacked per word on the host + rxgpu_ft_set_word_positions (eight uploads per word)
for a Zipf-shaped dictionary built by repeating a few hundred distinct streams under different word ids.  Usage:
  python tools/bench_ft_packed.py [--words 100000] [--out file.json]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def run(opts: dict) -> dict:
    class A:
        pass
    args = A()
    args.words = int(opts.get("words", 100_000))
    args.host_words = int(opts.get("host_words", 5_000))
    args.out = opts.get("out", "")
    from reindexer_amd import hostapi
    from tests.ft_pack import pack_postings
    from tests.test_bm25_oracle import make_pos_postings
    hostapi.lib()
    nf, total = 3, 5_000_000
    rng = np.random.default_rng(3)
    lens = np.unique(np.minimum(20_000, np.maximum(1, (3000.0 / np.arange(1, 301) ** 1.1).astype(int))))   # distinct list lengths, Zipf-ish
    distinct = []
    for n in lens:
        s = make_pos_postings(rng, total, nf, int(n), 1.0, array_fields=False, max_pos=4000)
        data, afp = pack_postings(s["doc"], s["pos_off"], s["fpos"])
        distinct.append((data, afp, int(n), int(s["pos_off"][-1])))
    # word w takes one of the distinct streams by a heavy-tailed draw: distinct[] ascends in length, most words get short lists
    idx = np.clip((rng.pareto(1.2, args.words) * 3).astype(int), 0, len(distinct) - 1)
    words = [(w, distinct[idx[w]][0], distinct[idx[w]][1]) for w in range(args.words)]
    nbytes = sum(int(w[1].shape[0]) for w in words)
    npost = sum(distinct[i][2] for i in idx)
    npos = sum(distinct[i][3] for i in idx)
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(np.ones((total, nf), np.float32), np.ones(nf, np.float32), np.zeros(total, np.uint8))
    m.set_words_packed(words[:1000])   # warm-up (module load, allocator)
    m.read_packed_stats()
    m.read_packed_wall()
    t0 = time.perf_counter()
    m.set_words_packed(words, host_from_bytes=1 << 40)
    harness_s = time.perf_counter() - t0   # with this harness's marshalling of 100 000 numpy arrays (list -> one blob) on top
    dev_s = m.read_packed_wall() / 1e3     # inside the library: gather into pinned memory, upload, both passes, dictionary entries
    count_ms, write_ms, bytes_in, bytes_out = m.read_packed_stats()
    # parity inside the bench: a sample of words (every distinct stream is hit) against the arrays the host decoder derives
    from tests.ft_pack import flat_entries
    checked = same = 0
    for w in list(range(0, args.words, max(1, args.words // 200))) + [args.words - 1]:
        got = m.get_word(w)
        data, afp = words[w][1], words[w][2]
        host = hostapi.ft_unpack(data, afp)
        eo, ef, et, e1, ro = flat_entries(host["doc"], host["pos_off"], host["fpos"])
        ok = (np.array_equal(got["doc"], host["doc"]) and np.array_equal(got["pos_off"], host["pos_off"]) and np.array_equal(got["fpos"], host["fpos"])
              and np.array_equal(got["ent_off"], eo) and np.array_equal(got["ent_field"], ef) and np.array_equal(got["ent_tf"], et)
              and np.array_equal(got["ent_first"], e1) and np.array_equal(got["range_off"], ro))
        checked += 1
        same += int(ok)
    chk = m.get_word(args.words - 1)
    m.close()
    # the thread-per-word kernels of round 2 on the same dictionary (device time only), for the record
    import os
    os.environ["RXGPU_FT_PACKED_THREAD"] = "1"
    try:
        mt = hostapi.GpuFtMerger(nf)
        mt.set_docs(np.ones((total, nf), np.float32), np.ones(nf, np.float32), np.zeros(total, np.uint8))
        mt.set_words_packed(words[:1000])
        mt.read_packed_stats()
        mt.set_words_packed(words, host_from_bytes=1 << 40)
        t_count_ms, t_write_ms, _, _ = mt.read_packed_stats()
        mt.close()
    finally:
        os.environ.pop("RXGPU_FT_PACKED_THREAD", None)
    m2 = hostapi.GpuFtMerger(nf)
    m2.set_docs(np.ones((total, nf), np.float32), np.ones(nf, np.float32), np.zeros(total, np.uint8))
    hw = words[:args.host_words]
    t0 = time.perf_counter()
    m2.set_words_packed(hw, host_from_bytes=0)   # every stream through AppendPacked + rxgpu_ft_set_word_positions
    host_s = time.perf_counter() - t0
    chk2 = m2.get_word(args.words - 1) if args.host_words >= args.words else None
    m2.close()
    hbytes = sum(int(w[1].shape[0]) for w in hw)
    out_bytes = npost * (4 + 4 + 4) + npos * 8 + npost * 9   # doc, pos_off, ent_off, positions, ~1 entry per posting
    res = {"workload": f"{args.words} dictionary words as PackedIdRelVec streams, {nbytes / 1e6:.1f} MB packed, {npost} postings, {npos} positions",
           "device": {"seconds": dev_s, "seconds_through_python_harness": harness_s, "seconds_is": "wall inside rxgpu_ft_set_words_packed_ptrs (C-ABI boundary)",
                      "words_per_sec": args.words / dev_s, "packed_MB_per_sec": nbytes / 1e6 / dev_s,
                      "postings_per_sec": npost / dev_s, "flat_bytes_written": out_bytes,
                      "kernels": {"form": "ft_packed_wave: one wavefront per word (256-byte windows, ballot varint ends, wave-uniform walk, LDS-staged "
                                          "coalesced output)",
                                  "count_ms": count_ms, "write_ms": write_ms, "stream_bytes_per_pass": bytes_in, "array_bytes_written": bytes_out,
                                  "thread_per_word_count_ms": t_count_ms, "thread_per_word_write_ms": t_write_ms,
                                  "speedup_vs_thread_per_word": (t_count_ms + t_write_ms) / max(count_ms + write_ms, 1e-9)},
                      "roofline": {"bound": "hbm", "kernel": "ft_packed_wave<write>", "unit": "GB/s", "peak": 8000.0,
                                   "bytes_per_launch": bytes_in + bytes_out, "avg_ms": write_ms,
                                   "achieved": (bytes_in + bytes_out) / max(write_ms, 1e-9) / 1e6,
                                   "frac": (bytes_in + bytes_out) / max(write_ms, 1e-9) / 1e6 / 8000.0,
                                   "flat_output_GBps_both_passes": bytes_out / max(count_ms + write_ms, 1e-9) / 1e6,
                                   "note": "bytes = packed streams read + flat arrays written by the write pass (the count pass reads the streams once more)"},
                      "parity": {"words_checked": checked, "identical_to_host_decoder_frac": same / max(checked, 1),
                                 "against": "PositionPostings::AppendPacked (pinned to the reference's packer) + the entry / range derivation of the tests"}},
           "host_path": {"words": args.host_words, "seconds": host_s, "words_per_sec": args.host_words / host_s,
                         "packed_MB_per_sec": hbytes / 1e6 / host_s},
           "speedup_words_per_sec": (args.words / dev_s) / (args.host_words / host_s),
           "last_word_postings": int(chk["doc"].shape[0])}
    if args.out:
        Path(args.out).write_text(json.dumps(res, indent=1))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--words", type=int, default=100_000)
    ap.add_argument("--host-words", type=int, default=5_000, help="words timed through the per-word host path (it is slow)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    print(json.dumps(run(dict(words=a.words, host_words=a.host_words, out=a.out))))


if __name__ == "__main__":
    main()
